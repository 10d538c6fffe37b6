// micro-benchmark: L2-resident load throughput of the access shapes of the A kernels on gfx950 (rows of 16-byte lanes): 128-byte row segments move data as fast as contiguous loads, 64-byte segments at half the rate (tuning aid)
#include <hip/hip_runtime.h>
#include <cstdio>
// each wave instruction: `rows` row segments of (64/rows) lanes x 16 bytes; rows are `pitch` bytes apart
__global__ void k(const float4* __restrict__ src, float* out, int iters, int rows, int pitch16) {
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int per = 64 / rows;
    const float4* p = src + (size_t)(lane / per) * pitch16 + (lane % per) + (wave & 15) * 64;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        unsigned o = (unsigned)it & 3u;
        asm volatile("" : "+v"(o));                          // opaque: the loads cannot be hoisted
#pragma unroll
        for (int u = 0; u < 8; ++u) { const float4 v = p[((size_t)u * rows * pitch16 + o * 8) % 4096]; acc += v.x + v.w; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    float4* src; float* out; hipMalloc(&src, 64 << 20); hipMalloc(&out, 256 * 16 * 256 * 4); hipMemset(src, 0, 64 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rows : {1, 2, 4, 8, 16, 32, 64}) {
        const int iters = 2000, blocks = 256 * 4, threads = 256, pitch16 = 240;          // 3840-byte rows (480 float2)
        k<<<blocks, threads>>>(src, out, 10, rows, pitch16);
        hipEventRecord(e0); k<<<blocks, threads>>>(src, out, iters, rows, pitch16); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)blocks * threads / 64 * iters * 8;
        printf("%2d rows x %2d lanes x 16 B per instruction: %.1f clk per wave instruction, %.1f B/clk/CU\n", rows, 64 / rows,
               ms * 1e-3 * 2.4e9 * 256 / instr, instr * 1024 / (ms * 1e-3 * 2.4e9 * 256));
    }
    return 0;
}
