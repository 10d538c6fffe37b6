// layout_probe.hip -- how fast do the two access patterns of the path move a half spectrum [241][640] float2 (1.23 MB, 256
// of them per launch) under different memory layouts?
//   A pattern: a workgroup owns 16 columns x all 241 rows (kA_* kernels): per row a 128-byte segment, 16 bytes per lane.
//   B pattern: a workgroup owns 5 whole rows (kB kernels): thread j touches columns j + 80 q, 8 bytes per lane.
// Layouts: 0 = row-major [k][640] (what the library uses);  1 = column groups of 64: [x/64][k][64];  2 = column tiles of 16:
// [x/16][k][16] (an A tile is one contiguous 30 KB block).
// Build: hipcc --offload-arch=gfx950 -O3 layout_probe.hip -o layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int HR = 241, W = 640;
template <int LAY> __device__ __forceinline__ size_t at(int k, int x) {
    if (LAY == 0) return (size_t)k * W + x;
    if (LAY == 1) return (size_t)(x >> 6) * (HR * 64) + (size_t)k * 64 + (x & 63);
    return (size_t)(x >> 4) * (HR * 16) + (size_t)k * 16 + (x & 15);
}
__device__ __forceinline__ void coords(int nbx, int n_items, int& bx, int& item) {      // an item's tiles share an XCD (as xcd_coords)
    const int L = blockIdx.x, xcd = L & 7, q = L >> 3;
    item = (q / nbx) * 8 + xcd; bx = q % nbx;
}
template <int LAY, bool WR> __global__ __launch_bounds__(256) void kA(float2* __restrict__ p, int n_items, float* out) {
    int bx, item; coords(W / 16, n_items, bx, item);
    float2* s = p + (size_t)item * HR * W;
    const int x0 = bx * 16, tid = threadIdx.x;
    float4 v[8]; float acc = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = tid + it * 256, k = idx >> 3, x2 = idx & 7;
        if (k < HR) {
            float4* g = reinterpret_cast<float4*>(s + at<LAY>(k, x0 + 2 * x2));
            if (WR) *g = make_float4((float)k, (float)x2, 1.f, 2.f); else v[it] = *g;
        } else v[it] = make_float4(0, 0, 0, 0);
    }
    if (!WR) {
#pragma unroll
        for (int it = 0; it < 8; ++it) acc += v[it].x + v[it].y + v[it].z + v[it].w;
        if (acc == 12345.f) out[0] = acc;
    }
}
template <int LAY, bool WR> __global__ __launch_bounds__(400) void kB(float2* __restrict__ p, int n_items, float* out) {
    const int item = blockIdx.y, lk = threadIdx.x / 80, j = threadIdx.x % 80, k = blockIdx.x * 5 + lk;
    if (k >= HR) return;
    float2* s = p + (size_t)item * HR * W;
    float2 v[8]; float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float2* g = s + at<LAY>(k, j + 80 * q);
        if (WR) *g = make_float2((float)k, (float)q); else v[q] = *g;
    }
    if (!WR) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q].x + v[q].y;
        if (acc == 12345.f) out[0] = acc;
    }
}
__global__ void k_flush(const float4* p, size_t n, float* out) {
    float s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x; }
    if (s == 12345.f) out[0] = s;
}
template <class F> double timeit(F f, float4* cold, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double tot = 0; const int REP = 8;
    for (int r = 0; r < REP + 1; ++r) {
        k_flush<<<4096, 256>>>(cold, ((size_t)1024 << 20) / 16, out);
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r) tot += ms;
    }
    return tot / REP;
}
int main() {
    const int N = 256;
    const size_t bytes = (size_t)N * HR * W * 8;
    float2* p; float4* cold; float* out;
    (void)hipMalloc(&p, bytes); (void)hipMalloc(&cold, (size_t)1024 << 20); (void)hipMalloc(&out, 4);
    hipMemset(p, 0, bytes); hipMemset(cold, 0, (size_t)1024 << 20);
    const dim3 gA(N * (W / 16)), gB((HR + 4) / 5, N);
    printf("%d spectra of %.2f MB; GB/s\nlayout   A-read   A-write   B-read   B-write\n", N, HR * W * 8 / 1e6);
#define ROW(L) { \
    const double ar = timeit([&] { kA<L, false><<<gA, 256>>>(p, N, out); }, cold, out); \
    const double aw = timeit([&] { kA<L, true><<<gA, 256>>>(p, N, out); }, cold, out); \
    const double br = timeit([&] { kB<L, false><<<gB, 400>>>(p, N, out); }, cold, out); \
    const double bw = timeit([&] { kB<L, true><<<gB, 400>>>(p, N, out); }, cold, out); \
    printf("%6d  %7.0f  %8.0f  %7.0f  %8.0f\n", L, bytes / ar / 1e6, bytes / aw / 1e6, bytes / br / 1e6, bytes / bw / 1e6); }
    ROW(0) ROW(1) ROW(2)
    return 0;
}
