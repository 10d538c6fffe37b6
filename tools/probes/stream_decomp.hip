// stream_decomp.hip -- where do the 20-35 % between a float4 copy (6.3 TB/s) and the memory-only time of the path's kernels
// (4.0-6.0 TB/s, DESIGN 4.2 "no FFT" column) go?  (fallback deliverable.)
//
// The probe moves the rotation stage's planes (256 items of [361][480] float2 = 1.386 MB each) with NO arithmetic, one step
// of realism at a time, and prints GB/s on the bytes it moves:
//   s0  grid-stride float4 copy (the guide's 6.29 TB/s figure), read-only, write-only
//   s1  the kernels' read : write MIXES on contiguous 16-byte lanes (2:3 product kernel, 3:2 ridge solve, 1:1, 1:0)
//   s2  the B kernels' tile: LK lines x T threads, RF strided 8-byte loads per thread (load_strided), one tile per workgroup,
//       as many workgroups per CU as fit
//   s3  s2 at the real kernels' occupancy (dynamic LDS padded to the exchange buffers' size: 4 / 5 workgroups per CU)
//   s4  the A kernels' tile: 12 / 16 columns x all rows, 16 bytes per lane over 96- / 128-byte row segments (a_load_pre,
//       a_post_store), read-only (arg-max) and read + write (kernel_fwd), free and real occupancy
// Build: hipcc --offload-arch=gfx950 -O3 stream_decomp.hip -o stream_decomp ; run: ./stream_decomp [items]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int HR = 361, N = 480;                 // rows x line length of the polar half spectrum
constexpr size_t PLANE = (size_t)HR * N;         // float2 per plane

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t nr, size_t nw, float* out) {
    float s = 0.f;
    const size_t stride = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = i0; i < (nr > nw ? nr : nw); i += stride) {
        float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
        if (i < nr) { v = a[i]; if (nw == 0) s += v.x; }
        if (i < nw) b[i] = v;
    }
    if (s == 12345.f) out[0] = s;
}
// contiguous 16-byte lanes, R read planes and W write planes per item, one "tile" of 5 lines per workgroup-iteration
template <int R, int W> __global__ __launch_bounds__(256) void k_mix(const float4* __restrict__ src, float4* __restrict__ dst, int n_items, float* out) {
    const size_t per = PLANE / 2;                 // float4 per plane
    float s = 0.f;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < per * n_items; t += (size_t)gridDim.x * 256) {
        const size_t item = t / per, o = t - item * per;
        float4 v = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) { const float4 u = src[(item * R + r) * per + o]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
#pragma unroll
        for (int w = 0; w < W; ++w) dst[(item * W + w) * per + o] = v;
        if (W == 0) s += v.x;
    }
    if (s == 12345.f) out[0] = s;
}
// B tile: LK lines x T threads; thread j of a line moves elements j + q*T, q < RF (8 bytes per lane, like load_strided)
template <int LK, int T, int RF, int R, int W> __global__ __launch_bounds__(LK * T) void k_btile(const float2* __restrict__ src, float2* __restrict__ dst, float* out) {
    extern __shared__ char pad[];
    const int item = blockIdx.y, lk = threadIdx.x / T, j = threadIdx.x % T, k = blockIdx.x * LK + lk;
    if (k >= HR) return;
    float2 v[R > 0 ? R : 1][RF];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < RF; ++q) v[r][q] = src[((size_t)item * R + r) * PLANE + (size_t)k * N + j + q * T];
    float2 acc[RF];
#pragma unroll
    for (int q = 0; q < RF; ++q) { acc[q] = make_float2((float)q, 1.f); for (int r = 0; r < R; ++r) { acc[q].x += v[r][q].x; acc[q].y += v[r][q].y; } }
#pragma unroll
    for (int w = 0; w < W; ++w)
#pragma unroll
        for (int q = 0; q < RF; ++q) dst[((size_t)item * W + w) * PLANE + (size_t)k * N + j + q * T] = acc[q];
    if (W == 0) { float s = 0; for (int q = 0; q < RF; ++q) s += acc[q].x; if (s == 12345.f) out[0] = s; }
    if (pad[0] == 77 && out[1] == 3.f) out[2] = 1.f;   // (keeps the dynamic LDS allocation alive)
}
// A tile: LX columns x all HR rows; 16 bytes per lane (two columns), LX/2 lanes per row segment
template <int LX, int NT, int R, int W> __global__ __launch_bounds__(NT) void k_atile(const float2* __restrict__ src, float2* __restrict__ dst, int n_items, float* out) {
    extern __shared__ char pad[];
    constexpr int LX2 = LX / 2, TOT = LX2 * HR, ITERS = (TOT + NT - 1) / NT, NBX = N / LX;
    const int L = blockIdx.x, xcd = L & 7, qq = L >> 3, item = (qq / NBX) * 8 + xcd, bx = qq % NBX;   // an item's tiles share an XCD
    if (item >= n_items) return;
    float4 v[ITERS]; float s = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int idx = threadIdx.x + it * NT;
        v[it] = make_float4(0, 0, 0, 0);
        if (idx < TOT) {
            const int x2 = idx % LX2, k = idx / LX2;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float4 u = *reinterpret_cast<const float4*>(src + ((size_t)item * R + r) * PLANE + (size_t)k * N + bx * LX + 2 * x2);
                v[it].x += u.x; v[it].y += u.y; v[it].z += u.z; v[it].w += u.w;
            }
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int idx = threadIdx.x + it * NT;
        if (idx < TOT) {
            const int x2 = idx % LX2, k = idx / LX2;
#pragma unroll
            for (int w = 0; w < W; ++w) *reinterpret_cast<float4*>(dst + ((size_t)item * W + w) * PLANE + (size_t)k * N + bx * LX + 2 * x2) = v[it];
            if (W == 0) s += v[it].x;
        }
    }
    if (s == 12345.f) out[0] = s;
    if (pad[0] == 77 && out[1] == 3.f) out[2] = 1.f;
}

static float4* g_cold; static size_t g_cold_n; static float* g_out;
template <class F> static double timeit(F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int rep = 0; rep < 7; ++rep) {
        hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, g_cold, g_cold, g_cold_n, (size_t)0, g_out);   // evict the Infinity Cache
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}
static void report(const char* name, double ms, double bytes) { printf("%-64s %8.4f ms  %7.0f GB/s\n", name, ms, bytes / (ms * 1e-3) / 1e9); fflush(stdout); }

int main(int argc, char** argv) {
    const int items = argc > 1 ? atoi(argv[1]) : 256;
    const size_t pb = PLANE * sizeof(float2);
    float2 *src, *dst;
    CK(hipMalloc(&src, pb * items * 3)); CK(hipMalloc(&dst, pb * items * 3)); CK(hipMalloc(&g_out, 64));
    g_cold_n = (size_t)512 << 20 >> 4; CK(hipMalloc(&g_cold, g_cold_n * 16));
    CK(hipMemset(src, 0, pb * items * 3)); CK(hipMemset(dst, 0, pb * items * 3)); CK(hipMemset(g_cold, 0, g_cold_n * 16)); CK(hipMemset(g_out, 0, 64));
    const double P = (double)pb * items;          // bytes of one plane over the batch
    printf("planes [%d][%d] float2, %d items (%.0f MB per plane-batch)\n", HR, N, items, P / 1e6);
    const size_t n4 = PLANE / 2 * items;
    report("s0 float4 copy 1:1 (grid-stride, 2048 x 256)", timeit([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const float4*)src, (float4*)dst, n4 * 2, n4 * 2, g_out); }), 4 * P);
    report("s0 float4 read only", timeit([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const float4*)src, (float4*)dst, n4 * 3, (size_t)0, g_out); }), 3 * P);
    report("s0 float4 write only", timeit([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const float4*)src, (float4*)dst, (size_t)0, n4 * 3, g_out); }), 3 * P);
    report("s1 mix 2 read : 3 write (fwd_mul_inv), 16 B lanes", timeit([&] { hipLaunchKernelGGL((k_mix<2, 3>), dim3(4096), dim3(256), 0, 0, (const float4*)src, (float4*)dst, items, g_out); }), 5 * P);
    report("s1 mix 2 read : 1 write (solve_inv)", timeit([&] { hipLaunchKernelGGL((k_mix<2, 1>), dim3(4096), dim3(256), 0, 0, (const float4*)src, (float4*)dst, items, g_out); }), 3 * P);
    report("s1 mix 2 read : 2 write (kernel_fwd)", timeit([&] { hipLaunchKernelGGL((k_mix<2, 2>), dim3(4096), dim3(256), 0, 0, (const float4*)src, (float4*)dst, items, g_out); }), 4 * P);
    report("s1 mix 1 read : 0 write (arg-max)", timeit([&] { hipLaunchKernelGGL((k_mix<1, 0>), dim3(4096), dim3(256), 0, 0, (const float4*)src, (float4*)dst, items, g_out); }), 1 * P);
#define BT(LK, T, RF, R, W, lds, label) { \
        if (lds > 65536) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_btile<LK, T, RF, R, W>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        report(label, timeit([&] { hipLaunchKernelGGL((k_btile<LK, T, RF, R, W>), dim3((HR + LK - 1) / LK, items), dim3(LK * T), lds, 0, src, dst, g_out); }), (double)(R + W) * P); }
    BT(5, 24, 20, 2, 3, 0, "s2 B tile 5 x 24 thr, 20 x 8 B strided, 2:3, free occupancy")
    BT(5, 24, 20, 2, 3, 40320, "s3 B tile 2:3 at 40 KB LDS (4 WG/CU = kB<480,fwd_mul_inv>)")
    BT(5, 24, 20, 2, 3, 80640, "s3 B tile 2:3 at 80 KB LDS (2 WG/CU = the ring form's consumers)")
    BT(8, 24, 20, 2, 1, 0, "s2 B tile 8 x 24 thr, 2:1, free occupancy")
    BT(8, 24, 20, 2, 1, 40000, "s3 B tile 2:1 at 40 KB LDS (4 WG/CU = kB<480,solve_inv> by VGPRs)")
    BT(5, 24, 20, 1, 0, 0, "s2 B tile read only")
    BT(5, 24, 20, 0, 1, 0, "s2 B tile write only")
#define AT(LX, NT, R, W, lds, label) { \
        report(label, timeit([&] { hipLaunchKernelGGL((k_atile<LX, NT, R, W>), dim3((N / LX) * ((items + 7) / 8 * 8)), dim3(NT), lds, 0, src, dst, items, g_out); }), (double)(R + W) * P); }
    AT(12, 240, 2, 2, 0, "s4 A tile 12 col x 361 rows, 16 B lanes / 96 B segments, 2:2, free occ.")
    AT(12, 240, 2, 2, 35000, "s4 A tile 2:2 at 35 KB LDS (4 WG/CU = kA_inv<360,kernel_fwd>)")
    AT(12, 240, 1, 0, 0, "s4 A tile read only (arg-max), free occupancy")
    AT(12, 240, 1, 0, 35000, "s4 A tile read only at 35 KB LDS (4 WG/CU = kA_inv<360,argmax>)")
    AT(16, 256, 2, 2, 0, "s4 A tile 16 col (128 B segments), 2:2, free occupancy")
    AT(16, 256, 1, 0, 0, "s4 A tile 16 col read only, free occupancy")
    AT(12, 240, 0, 1, 0, "s4 A tile write only")
    return 0;
}
