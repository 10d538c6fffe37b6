// placement_probe.hip -- does WHERE the work buffers land in HBM decide the rate of the HBM-bound B kernels?
// (round 6: re-creating the context moves kB<480,fwd_mul_inv> between 0.555 and 0.615 ms per 512 items, and physically
// contiguous buffers -- hipDeviceMallocContiguous -- make it 0.69-0.83: profiles/r06_placement_probe.txt, r06_placement_contig.txt)
//
// The kernel below has kB<480,fwd_mul_inv>'s data movement and nothing else: a workgroup owns 5 rows of one item's [361][480]
// complex plane, reads them from two planes (X: item i of buffer A, Z: item i of buffer B) and writes them to three (buffers C, D, E);
// 512 items, item-major blocks.  It is timed on buffers obtained four ways:
//   malloc      five hipMallocs (what the library does)
//   contiguous  hipExtMallocWithFlags(hipDeviceMallocContiguous)
//   vmm-seq     hipMemAddressReserve + one hipMemCreate per CHUNK, mapped in creation order
//   vmm-perm    the same chunks, mapped in a fixed pseudo-random order over all five buffers (neighbouring items of one buffer
//               sit in chunks that were created far apart)
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/placement_probe.hip -o tools/probes/placement_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ROWS = 361, COLS = 480, LK = 5, T = 24, PTS = COLS / T;      // 20 points per thread, as the real kernel
constexpr size_t PLANE = (size_t)ROWS * COLS;                                // float2 elements per item

__global__ __launch_bounds__(LK * T) void k_move(const float2* __restrict__ A, const float2* __restrict__ B, float2* __restrict__ C,
                                                 float2* __restrict__ D, float2* __restrict__ E, int tiles) {
    const int item = blockIdx.x / tiles, tile = blockIdx.x - item * tiles;
    const int line = threadIdx.x / T, j = threadIdx.x - line * T;
    const int row = tile * LK + line;
    if (row >= ROWS) return;
    const size_t off = (size_t)item * PLANE + (size_t)row * COLS + j;
    float2 x[PTS], z[PTS];
#pragma unroll
    for (int q = 0; q < PTS; ++q) { x[q] = A[off + q * T]; z[q] = B[off + q * T]; }
#pragma unroll
    for (int q = 0; q < PTS; ++q) {
        C[off + q * T] = x[q];
        D[off + q * T] = make_float2(z[q].x * z[q].x + z[q].y * z[q].y, 0.f);
        E[off + q * T] = make_float2(x[q].x * z[q].x + x[q].y * z[q].y, x[q].y * z[q].x - x[q].x * z[q].y);
    }
}

static double time_kernel(float2* const p[5], int items, int reps) {
    const int tiles = (ROWS + LK - 1) / LK;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_move, dim3(items * tiles), dim3(LK * T), 0, 0, p[0], p[1], p[2], p[3], p[4], tiles);
    hipEventRecord(a, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_move, dim3(items * tiles), dim3(LK * T), 0, 0, p[0], p[1], p[2], p[3], p[4], tiles);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / reps;
}

int main(int argc, char** argv) {
    const int items = argc > 1 ? atoi(argv[1]) : 512, rounds = argc > 2 ? atoi(argv[2]) : 3;
    const size_t chunk_mb = argc > 3 ? (size_t)atoi(argv[3]) : 2;
    const size_t bytes = PLANE * sizeof(float2) * items;
    const double moved = 5.0 * bytes;
    printf("items %d, %.2f GB per buffer, 5 buffers, %.2f GB moved per launch; VMM chunk %zu MiB\n", items, bytes / 1e9, moved / 1e9, chunk_mb);
    for (int rd = 0; rd < rounds; ++rd) {
        {   // malloc
            float2* p[5];
            for (auto& q : p) CK(hipMalloc(&q, bytes));
            for (auto& q : p) CK(hipMemset(q, 0, bytes));
            const double ms = time_kernel(p, items, 20);
            printf("round %d  malloc      %.4f ms  %.0f GB/s\n", rd, ms, moved / ms / 1e6);
            for (auto& q : p) hipFree(q);
        }
        {   // contiguous
            float2* p[5]; bool ok = true;
            for (auto& q : p) { q = nullptr; if (hipExtMallocWithFlags((void**)&q, bytes, hipDeviceMallocContiguous) != hipSuccess) { ok = false; (void)hipGetLastError(); } }
            if (ok) {
                for (auto& q : p) CK(hipMemset(q, 0, bytes));
                const double ms = time_kernel(p, items, 20);
                printf("round %d  contiguous  %.4f ms  %.0f GB/s\n", rd, ms, moved / ms / 1e6);
            } else printf("round %d  contiguous  allocation failed\n", rd);
            for (auto& q : p) if (q) hipFree(q);
        }
        for (int perm = 0; perm < 2; ++perm) {   // VMM
            hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
            const size_t chunk = std::max(gran, chunk_mb << 20);
            const size_t per = (bytes + chunk - 1) / chunk, total = 5 * per;
            std::vector<hipMemGenericAllocationHandle_t> h(total);
            for (size_t i = 0; i < total; ++i) CK(hipMemCreate(&h[i], chunk, &prop, 0));
            std::vector<size_t> order(total);
            std::iota(order.begin(), order.end(), 0);
            if (perm) { size_t s = 12345; for (size_t i = total - 1; i > 0; --i) { s = s * 6364136223846793005ull + 1442695040888963407ull; std::swap(order[i], order[(s >> 33) % (i + 1)]); } }
            void* base[5]; float2* p[5];
            hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            for (int b = 0; b < 5; ++b) {
                CK(hipMemAddressReserve(&base[b], per * chunk, 0, nullptr, 0));
                for (size_t k = 0; k < per; ++k) CK(hipMemMap((char*)base[b] + k * chunk, chunk, 0, h[order[b * per + k]], 0));
                CK(hipMemSetAccess(base[b], per * chunk, &acc, 1));
                p[b] = (float2*)base[b];
                CK(hipMemset(p[b], 0, bytes));
            }
            const double ms = time_kernel(p, items, 20);
            printf("round %d  vmm-%s    %.4f ms  %.0f GB/s   (granularity %zu KiB, %zu chunks)\n", rd, perm ? "perm" : "seq ", ms, moved / ms / 1e6, gran >> 10, total);
            for (int b = 0; b < 5; ++b) { hipMemUnmap(base[b], per * chunk); hipMemAddressFree(base[b], per * chunk); }
            for (auto& x : h) hipMemRelease(x);
        }
    }
    return 0;
}
