// overlap_probe.hip -- can a bandwidth-bound kernel and an FP32-bound kernel share the chip without slowing each other?
// M = streaming copy (1 GiB read + 1 GiB write), C = FMA chains with no memory traffic; M alone, C alone, both on two streams.
// Build: hipcc --offload-arch=gfx950 -O3 overlap_probe.hip -o overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_copy(const float4* a, float4* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_fma(float* out, int iters, float s) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
        a0 = fmaf(a0, s, 1.f); a1 = fmaf(a1, s, 1.f); a2 = fmaf(a2, s, 1.f); a3 = fmaf(a3, s, 1.f);
        a4 = fmaf(a4, s, 1.f); a5 = fmaf(a5, s, 1.f); a6 = fmaf(a6, s, 1.f); a7 = fmaf(a7, s, 1.f);
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (r == 12345.f) out[0] = r;
}
// the same work in ONE kernel: every thread alternates a strip of copy with a strip of FMAs (what a fused FFT kernel does)
__global__ void k_both(const float4* a, float4* b, size_t n, float* out, int iters, float s) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        float4 v = a[i];
        for (int k = 0; k < iters; ++k) { a0 = fmaf(a0, s, v.x); a1 = fmaf(a1, s, v.y); a2 = fmaf(a2, s, v.z); a3 = fmaf(a3, s, v.w); }
        v.x += a0; v.y += a1; v.z += a2; v.w += a3;
        b[i] = v;
    }
    if (a0 == 12345.f) out[0] = a0;
}
int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    float4 *a, *b; float* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    const int GM = 256 * 8, GC = 256 * 8, T = 256;
    auto ms = [&](hipEvent_t x, hipEvent_t y) { float t; (void)hipEventElapsedTime(&t, x, y); return t; };
    for (int iters : { 2000, 4000, 8000 }) {
        float tm = 0, tc = 0, tb_m = 0, tb_c = 0, wall = 0;
        for (int r = 0; r < 6; ++r) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s1)); k_copy<<<GM, T, 0, s1>>>(a, b, n); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
            float m = ms(e0, e1);
            CK(hipEventRecord(f0, s2)); k_fma<<<GC, T, 0, s2>>>(out, iters, 0.5f); CK(hipEventRecord(f1, s2)); CK(hipDeviceSynchronize());
            float c = ms(f0, f1);
            CK(hipEventRecord(e0, s1)); CK(hipEventRecord(f0, s2));
            k_copy<<<GM, T, 0, s1>>>(a, b, n); k_fma<<<GC, T, 0, s2>>>(out, iters, 0.5f);
            CK(hipEventRecord(e1, s1)); CK(hipEventRecord(f1, s2)); CK(hipDeviceSynchronize());
            if (r) { tm += m; tc += c; tb_m += ms(e0, e1); tb_c += ms(f0, f1); }
        }
        printf("fma iters %5d: copy alone %.3f ms (%.0f GB/s)  fma alone %.3f ms   together: copy %.3f ms, fma %.3f ms\n", iters, tm / 5, 2.0 * bytes / (tm / 5 * 1e-3) / 1e9, tc / 5, tb_m / 5, tb_c / 5);
    }
    for (int iters : { 0, 8, 16, 32, 64 }) {
        float t = 0;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0, s1)); k_both<<<GM, T, 0, s1>>>(a, b, n, out, iters, 0.5f); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
            if (r) t += ms(e0, e1);
        }
        printf("fused copy + %2d x 4 fma per 16 bytes: %.3f ms (%.0f GB/s, %.1f TFLOP/s)\n", iters, t / 5, 2.0 * bytes / (t / 5 * 1e-3) / 1e9, 8.0 * iters * n / (t / 5 * 1e-3) / 1e12);
    }
    return 0;
}
