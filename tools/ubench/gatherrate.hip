// micro-benchmark: L1-resident gather rate of gfx950 for 4-byte and (4-byte aligned) 8-byte loads (tuning aid)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f2a __attribute__((ext_vector_type(2)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
// MODE 0: dword, random within `span` floats; 1: unaligned dwordx2 random; 2: aligned dwordx2 random (even index);
// SPREAD: lanes of a wave fall into `lines` different 128-byte lines (rest of the address random inside the line)
template <int MODE> __global__ void k(const float* __restrict__ src, float* out, int iters, unsigned span_mask, int lines) {
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const unsigned lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            unsigned line = (lines >= 64) ? ((s >> 8) & (span_mask >> 5)) : ((lane % lines) + ((s >> 20) & 3) * lines);
            unsigned idx = (line << 5) | ((s >> 3) & 31);
            idx &= span_mask;
            if (MODE == 0) acc += src[idx];
            else if (MODE == 1) { idx = idx > span_mask - 1 ? span_mask - 1 : idx; const f2u v = *reinterpret_cast<const f2u*>(src + idx); acc += v.x + v.y; }
            else if (MODE == 2) { idx &= ~1u; const f2a v = *reinterpret_cast<const f2a*>(src + idx); acc += v.x + v.y; }
            else { idx = idx > span_mask - 3 ? span_mask - 3 : idx; const f4u v = *reinterpret_cast<const f4u*>(src + idx); acc += v.x + v.y + v.z + v.w; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE> void run(const char* name, unsigned span_floats, int lines) {
    float *src, *out; hipMalloc(&src, 1 << 24); hipMalloc(&out, 256 * 16 * 256 * 4); hipMemset(src, 0, 1 << 24);
    const int iters = 2000, blocks = 256 * 4, threads = 256;          // 16 waves per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(src, out, 10, span_floats - 1, lines);
    hipEventRecord(e0); k<MODE><<<blocks, threads>>>(src, out, iters, span_floats - 1, lines); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double lane_loads = (double)blocks * threads * iters * 8;
    printf("%-22s span %7u floats, %2d lines/instr: %.2f lane-loads/clk/CU  (%.1f clk per wave instruction)\n", name, span_floats, lines,
           lane_loads / (ms * 1e-3 * 2.4e9 * 256), 64.0 / (lane_loads / (ms * 1e-3 * 2.4e9 * 256)));
    hipFree(src); hipFree(out);
}
int main() {
    for (int lines : {1, 2, 4, 8, 16, 32, 64}) {
        run<0>("dword", 4096, lines); run<1>("dwordx2 align 4", 4096, lines); run<2>("dwordx2 align 8", 4096, lines); run<3>("dwordx4 align 4", 4096, lines);
    }
    for (unsigned span : {1u << 14, 1u << 17, 1u << 20}) { run<1>("dwordx2 align 4", span, 64); run<3>("dwordx4 align 4", span, 64); }
    return 0;
}
