// mall_probe.hip -- does a buffer written by one kernel come back from the 256 MiB Infinity Cache when the next kernel
// reads it?  write S MiB, then read S MiB (sum), for S = 16 ... 2048; GB/s of the write, the read-after-write and a
// read of a cold buffer of the same size.  Build: hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_write(float4* p, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(v, v, v, v);
}
__global__ void k_read(const float4* p, size_t n, float* out) {
    float s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) out[0] = s;
}
__global__ void k_copy(const float4* a, float4* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    const size_t MAXB = (size_t)2048 << 20;
    float4 *a, *b, *cold; float* out;
    (void)hipMalloc(&a, MAXB); (void)hipMalloc(&b, MAXB); (void)hipMalloc(&cold, MAXB); (void)hipMalloc(&out, 4);
    hipMemset(cold, 0, MAXB); hipMemset(a, 0, MAXB); hipMemset(b, 0, MAXB);
    hipEvent_t e0, e1, e2, e3; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
    const int G = 256 * 16, T = 256;
    for (size_t mb : { 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048 }) {
        const size_t bytes = mb << 20, n = bytes / 16;
        double tw = 0, tr = 0, tc = 0, tcp = 0; const int REP = 10;
        for (int r = 0; r < REP + 2; ++r) {
            k_read<<<G, T>>>(cold, MAXB / 16, out);          // flush the caches with 2 GiB of something else
            hipEventRecord(e0); k_write<<<G, T>>>(a, n, (float)r); hipEventRecord(e1);
            k_read<<<G, T>>>(a, n, out); hipEventRecord(e2);
            k_copy<<<G, T>>>(a, b, n); hipEventRecord(e3);
            hipEventSynchronize(e3);
            float w, rd, cp; hipEventElapsedTime(&w, e0, e1); hipEventElapsedTime(&rd, e1, e2); hipEventElapsedTime(&cp, e2, e3);
            // cold read of the same size
            k_read<<<G, T>>>(b + n, (MAXB - bytes) / 16 > n ? n : 0, out);
            hipEventRecord(e0); k_read<<<G, T>>>(cold, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
            float c; hipEventElapsedTime(&c, e0, e1);
            if (r >= 2) { tw += w; tr += rd; tc += c; tcp += cp; }
        }
        printf("%5zu MiB: write %7.0f GB/s   read-after-write %7.0f GB/s   copy(after) %7.0f GB/s(r+w)   read cold-ish %7.0f GB/s\n", mb,
               bytes / (tw / REP * 1e-3) / 1e9, bytes / (tr / REP * 1e-3) / 1e9, 2.0 * bytes / (tcp / REP * 1e-3) / 1e9, bytes / (tc / REP * 1e-3) / 1e9);
    }
    return 0;
}
