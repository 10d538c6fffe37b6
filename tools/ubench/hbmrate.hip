// micro-benchmark: practical HBM bandwidth of gfx950 for the access shapes the KCC kernels use (tuning aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_read(const float4* __restrict__ a, float* out, size_t n) {
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) out[0] = s;
}
__global__ void k_write(float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = make_float4(1, 2, 3, 4);
}
// one block per "tile": reads TILE bytes contiguous, writes TILE bytes contiguous (like a B kernel line group)
template <int PER> __global__ void k_tile(const float4* __restrict__ a, float4* __restrict__ b) {
    const size_t base = (size_t)blockIdx.x * blockDim.x * PER + threadIdx.x;
    float4 v[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) v[q] = a[base + (size_t)q * blockDim.x];
#pragma unroll
    for (int q = 0; q < PER; ++q) b[base + (size_t)q * blockDim.x] = v[q];
}
// transposed 64/128-byte row segments (A kernel shape): rows of `pitch` float4, each block moves SEG float4 of every row of an image
template <int SEG> __global__ void k_seg(const float4* __restrict__ a, float4* __restrict__ b, int rows, int pitch4) {
    const int img = blockIdx.y, seg = blockIdx.x;
    const float4* A = a + (size_t)img * rows * pitch4 + (size_t)seg * SEG; float4* B = b + (size_t)img * rows * pitch4 + (size_t)seg * SEG;
    for (int i = threadIdx.x; i < rows * SEG; i += blockDim.x) { const int r = i / SEG, c = i % SEG; B[(size_t)r * pitch4 + c] = A[(size_t)r * pitch4 + c]; }
}
int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    float4 *a, *b; float* o; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, double moved, auto f) {
        f(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms, moved / ms / 1e9);
    };
    for (int blocks : {2048, 8192, 32768}) {
        char nm[64];
        snprintf(nm, 64, "copy  grid-stride %d x 256", blocks); time(nm, 2.0 * bytes, [&] { k_copy<<<blocks, 256>>>(a, b, n); });
        snprintf(nm, 64, "read  grid-stride %d x 256", blocks); time(nm, 1.0 * bytes, [&] { k_read<<<blocks, 256>>>(a, o, n); });
        snprintf(nm, 64, "write grid-stride %d x 256", blocks); time(nm, 1.0 * bytes, [&] { k_write<<<blocks, 256>>>(b, n); });
    }
    time("tile copy 256 thr x 4 float4 (16 KB/block)", 2.0 * bytes, [&] { k_tile<4><<<(unsigned)(n / (256 * 4)), 256>>>(a, b); });
    time("tile copy 256 thr x 8 float4 (32 KB/block)", 2.0 * bytes, [&] { k_tile<8><<<(unsigned)(n / (256 * 8)), 256>>>(a, b); });
    time("tile copy 192 thr x 10 float4 (30 KB/block)", 2.0 * (n / 1920 * 1920) * 16, [&] { k_tile<10><<<(unsigned)(n / (192 * 10)), 192>>>(a, b); });
    // A-kernel shape: images of 361 rows x 480 float2 (= 240 float4 per row); 64-byte (4 float4) and 128-byte (8 float4) segments
    const int rows = 361, pitch4 = 240; const int imgs = (int)(n / ((size_t)rows * pitch4));
    time("row segments 64 B  (361 rows, 160 thr)", 2.0 * imgs * rows * pitch4 * 16, [&] { k_seg<4><<<dim3(pitch4 / 4, imgs), 160>>>(a, b, rows, pitch4); });
    time("row segments 128 B (361 rows, 256 thr)", 2.0 * imgs * rows * pitch4 * 16, [&] { k_seg<8><<<dim3(pitch4 / 8, imgs), 256>>>(a, b, rows, pitch4); });
    time("row segments 256 B (361 rows, 256 thr)", 2.0 * imgs * rows * pitch4 * 16, [&] { k_seg<16><<<dim3(pitch4 / 16, imgs), 256>>>(a, b, rows, pitch4); });
    return 0;
}
