// micro-benchmark: issue rate of scalar vs packed FP32 VALU ops on gfx950 (tuning aid, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(float* out, int iters) {
    v2 a[8]; float s[16];
    for (int i = 0; i < 8; ++i) { a[i].x = threadIdx.x * 0.001f + i; a[i].y = i * 0.5f; }
    for (int i = 0; i < 16; ++i) s[i] = threadIdx.x * 0.002f + i;
    v2 w; w.x = 0.999f; w.y = 0.001f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // 16 independent scalar FMAs
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(w.x), "v"(w.y));
        } else if (MODE == 1) {     // 8 independent packed FMAs (same flops)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(w));
        } else if (MODE == 2) {     // 8 packed adds with swap + neg modifiers
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(w));
        } else if (MODE == 3) {     // 16 scalar adds
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(w.x));
        } else if (MODE == 4) {     // 8 packed muls
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
        } else if (MODE == 5) {     // dependent chain packed fma (latency)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(w), "v"(w));
        } else if (MODE == 6) {     // dependent chain scalar fma
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[0]) : "v"(w.x), "v"(w.y));
        }
    }
    float r = 0; for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y; for (int i = 0; i < 16; ++i) r += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(const char* name, int instr_per_iter, int waves_per_simd) {
    float* d; hipMalloc(&d, 256 * 1024 * 4 * 8);
    const int iters = 20000, blocks = 256 * waves_per_simd, threads = 256;      // waves_per_simd waves on each of the 4 SIMDs of every CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(d, 10);
    hipEventRecord(e0); k<MODE><<<blocks, threads>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // instructions issued per SIMD: waves_per_simd * iters * instr_per_iter
    double per = ms * 1e-3 / ((double)waves_per_simd * iters * instr_per_iter);
    printf("%-28s waves/simd %d: %.3f ns per wave-instruction (%.2f clk @2.4GHz)\n", name, waves_per_simd, per * 1e9, per * 2.4e9);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32 x16", 16, w); run<1>("v_pk_fma_f32 x8", 8, w); run<2>("v_pk_add_f32 swapneg x8", 8, w);
        run<3>("v_add_f32 x16", 16, w); run<4>("v_pk_mul_f32 x8", 8, w); run<5>("v_pk_fma dep chain x8", 8, w); run<6>("v_fma dep chain x8", 8, w);
    }
    return 0;
}
