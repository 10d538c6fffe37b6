// xcd_pipeline_probe.hip -- go / no-go probe for XCD-resident stage pipelines (VERDICT r3 "next round" #2).
//
// Question: EstimateTrans (reference src/correlation_flow.cc:145-179) is four kernels whose intermediates (two kernel planes,
// 2.08 MB per 720x480 polar item) make a round trip through the fabric at every kernel boundary.  If the workgroups of ONE XCD
// walked an item through all four phases back to back, those re-reads could be served by that XCD's 4 MiB L2.  This probe
// reproduces the rotation stage's data movement -- the same planes, tile shapes, access granularities (8-byte lanes over
// contiguous spectrum rows in the B phases, 16-byte lanes over 96-byte row segments in the A phases), LDS footprint and a
// calibrated amount of dummy arithmetic per element -- in two forms:
//   separate : four launches per batch, as the library runs today;
//   fused    : ONE persistent launch of phase-specialised workgroups.  Every workgroup reads its XCC_ID; the k-th one to arrive
//              on an XCD takes a role (phase) from a 16-entry pattern and pulls tiles of that phase from the XCD's own counter
//              in item order (item i belongs to XCD i % 8).  A tile of phase p waits until all tiles of phase p-1 of the same
//              item have signalled (a per-item counter); phase 1 of item j waits for phase 4 of item j - W, so W items are in
//              flight per XCD -- what has to stay in its 4 MiB L2.  Every wait is for work that another, always resident, role
//              pulls without waiting on the waiter: no deadlock whatever the dispatcher does.
// Hand-off inside an XCD: producer = plain stores, s_waitcnt vmcnt(0) in every storing wave, workgroup barrier, one counter
// increment; consumer = one poller lane, then loads that bypass the CU's L1 (buffer_load ... sc1), served by the XCD's L2.
// Both sides of an item run on the XCD that owns the item by construction (a workgroup only pulls items of its own XCC_ID), so the protocol does not depend on where the dispatcher puts a block.  EVERY word handed over is checked against a
// tag that changes per launch (stale lines of the previous launch would be seen).
//
// Output: time per batch of both forms, number of mismatching words, census of workgroups per XCD, polls per ticket.
// rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this binary gives the fabric bytes of each kernel (gfx950: FETCH_SIZE counts
// 64 B per 128-byte line).
//
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_pipeline_probe xcd_pipeline_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NK = 361, NL = 480;                 // half-spectrum rows x columns of the 720 x 480 polar plane (complex)
constexpr int PLANE = NK * NL;                    // float2 elements per plane
constexpr int ZZC = 252;                          // stored columns of the Hermitian-half zz plane (kcc: zz_half_columns)
#ifndef PROBE_LK1
#define PROBE_LK1 5
#endif
#ifndef PROBE_LK3
#define PROBE_LK3 6
#endif
#ifndef PROBE_LXA
#define PROBE_LXA 12
#endif
constexpr int LK1 = PROBE_LK1, LK3 = PROBE_LK3, LXA = PROBE_LXA;   // rows per B tile (phase 1 / 3), columns per A tile (phases 2 / 4); -D overrides: tile-size study
constexpr int T1 = (NK + LK1 - 1) / LK1;          // 73 tiles
constexpr int T2Z = ZZC / LXA, T2X = NL / LXA;    // 21 + 40 tiles
constexpr int T2 = T2Z + T2X;
constexpr int T3 = (NK + LK3 - 1) / LK3;          // 61 tiles (the library: 8 rows, 46 tiles; 6 keeps every role under 96 VGPRs)
constexpr int T4 = NL / LXA;                      // 40 tiles
constexpr int NT = 256;
static int LDS_BYTES = 32 * 1024;                 // what the real kernels hold per workgroup (4-5 workgroups per CU); --lds overrides (occupancy study)

struct Params {
    const float2* X; const float2* Z;             // [items][PLANE] inputs (streamed once)
    float2* S;                                    // [items][2][PLANE] scratch: plane 0 = zz (columns < ZZC used), plane 1 = xz -> G
    int n_items;
    int f1, f2, f3, f4;                           // dummy FMAs per loaded complex element, per phase
    unsigned epoch;                               // tag salt of this launch
    int l1_bypass;                                // consumer loads: 1 = sc1 (bypass L1), 0 = plain
    unsigned* err;                                // mismatching words
    float* sink;
};

__device__ __forceinline__ unsigned tag_of(unsigned epoch, int item, int phase, int plane, int idx) {
    return (unsigned)idx * 2654435761u ^ (epoch * 0x9E3779B9u) ^ ((unsigned)item << 8) ^ ((unsigned)phase << 28) ^ ((unsigned)plane << 27);
}
// f dependent FMAs on each of the N values of a thread (the chains of different values interleave: full issue rate)
template <int N> __device__ __forceinline__ float burn(float (&v)[N], int f) {
    for (int i = 0; i < f; ++i) {
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = __builtin_fmaf(v[q], 0.999f, 0.001f);
    }
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < N; ++q) a += v[q];
    return a;
}
// 8-byte / 16-byte loads with or without the L1 bypass
__device__ __forceinline__ u32x2 ld8(const float2* base, __amdgpu_buffer_rsrc_t r, unsigned elem, int bypass) {
    if (bypass) return __builtin_amdgcn_raw_buffer_load_b64(r, elem * 8u, 0, 16);
    const float2 v = base[elem]; u32x2 o; o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y); return o;
}
__device__ __forceinline__ u32x4 ld16(const float2* base, __amdgpu_buffer_rsrc_t r, unsigned elem, int bypass) {
    if (bypass) return __builtin_amdgcn_raw_buffer_load_b128(r, elem * 8u, 0, 16);
    const float4 v = *reinterpret_cast<const float4*>(base + elem);
    u32x4 o; o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y); o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w); return o;
}

// ---- phase bodies (all 256 threads of the workgroup call them) ----------------------------------------------------
// phase 1 (kB fwd_mul_inv): rows [k0, k0+LK1) of X and Z in (streamed, non-temporal), zz (half) and xz rows out
__device__ __forceinline__ void phase1(const Params& p, int item, int tile, float* lds, int tid) {
    const int k0 = tile * LK1, rows = min(LK1, NK - k0), n = rows * NL;
    const float2* X = p.X + (size_t)item * PLANE + (size_t)k0 * NL;
    const float2* Z = p.Z + (size_t)item * PLANE + (size_t)k0 * NL;
    float2* zz = p.S + (size_t)item * 2 * PLANE + (size_t)k0 * NL;
    float2* xz = zz + PLANE;
    constexpr int PER = (LK1 * NL + NT - 1) / NT;
    float2 vx[PER], vz[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = tid + q * NT;
        if (e < n) {
            const f32x2 a = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(X + e)), b = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(Z + e));
            vx[q] = make_float2(a.x, a.y); vz[q] = make_float2(b.x, b.y);
        }
        else { vx[q] = make_float2(0, 0); vz[q] = vx[q]; }
    }
    float w[2 * PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) { w[2 * q] = vx[q].x + vz[q].y; w[2 * q + 1] = vx[q].y - vz[q].x; }
    const float acc = burn(w, p.f1);
    lds[tid] = acc;                                   // (keeps the LDS allocation alive)
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = tid + q * NT;
        if (e < n) {
            const int idx = k0 * NL + e, col = e % NL;
            xz[e] = make_float2(__uint_as_float(tag_of(p.epoch, item, 1, 1, idx)), acc);
            if (col < ZZC) zz[e] = make_float2(__uint_as_float(tag_of(p.epoch, item, 1, 0, idx)), acc);
        }
    }
}
// phases 2 and 4 (kA_inv kernel_fwd / argmax): 12 columns x all rows, 96-byte row segments, 16 bytes per lane;
// phase 2 rewrites the tile in place (the kernel plane's spectrum), phase 4 only reads
template <int PH>
__device__ __forceinline__ void phaseA(const Params& p, int item, int tile, float* lds, int tid) {
    int plane, x0;
    if (PH == 2) { plane = tile < T2Z ? 0 : 1; x0 = (plane ? tile - T2Z : tile) * LXA; }
    else { plane = 1; x0 = tile * LXA; }
    float2* base = p.S + (size_t)item * 2 * PLANE + (size_t)plane * PLANE;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, PLANE * 8, 0x00020000);
    constexpr int TOT = NK * (LXA / 2), PER = (TOT + NT - 1) / NT;
    u32x4 v[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int i = tid + q * NT;
        if (i < TOT) { const int k = i / (LXA / 2), x2 = i % (LXA / 2); v[q] = ld16(base, r, (unsigned)(k * NL + x0 + 2 * x2), p.l1_bypass); }
        else v[q] = (u32x4)(0u);
    }
    const int src_phase = PH == 2 ? 1 : 3;
    unsigned bad = 0; float w[2 * PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int i = tid + q * NT;
        w[2 * q] = __uint_as_float(v[q].y) * 0.f; w[2 * q + 1] = __uint_as_float(v[q].w) * 0.f;
        if (i < TOT) {
            const int k = i / (LXA / 2), x2 = i % (LXA / 2), idx = k * NL + x0 + 2 * x2;
            bad += (v[q].x != tag_of(p.epoch, item, src_phase, plane, idx)) + (v[q].z != tag_of(p.epoch, item, src_phase, plane, idx + 1));
        }
    }
    const float acc = burn(w, PH == 2 ? p.f2 : p.f4);
    if (bad) atomicAdd(p.err, bad);
    lds[tid] = acc;
    if (PH == 2) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = tid + q * NT;
            if (i < TOT) {
                const int k = i / (LXA / 2), x2 = i % (LXA / 2), idx = k * NL + x0 + 2 * x2;
                *reinterpret_cast<float4*>(base + idx) = make_float4(__uint_as_float(tag_of(p.epoch, item, 2, plane, idx)), acc,
                                                                     __uint_as_float(tag_of(p.epoch, item, 2, plane, idx + 1)), acc);
            }
        }
    } else if (acc == 123.456f) p.sink[0] = acc;
}
// phase 3 (kB solve_inv): rows of zz (stored half) and xz in, G rows out (over xz, in place)
__device__ __forceinline__ void phase3(const Params& p, int item, int tile, float* lds, int tid) {
    const int k0 = tile * LK3, rows = min(LK3, NK - k0), n = rows * NL;
    float2* zz = p.S + (size_t)item * 2 * PLANE;
    float2* xz = zz + PLANE;
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)zz, 0, PLANE * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xz, 0, PLANE * 8, 0x00020000);
    constexpr int PER = (LK3 * NL + NT - 1) / NT;
    u32x2 vz[PER], vx[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = tid + q * NT;
        if (e < n) {
            const int row = e / NL, col = e % NL, mc = col < ZZC ? col : NL - col;       // Hermitian mirror of the stored half
            vz[q] = ld8(zz, rz, (unsigned)((k0 + row) * NL + mc), p.l1_bypass);
            vx[q] = ld8(xz, rx, (unsigned)(k0 * NL + e), p.l1_bypass);
        } else { vz[q] = (u32x2)(0u); vx[q] = vz[q]; }
    }
    unsigned bad = 0; float w[2 * PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = tid + q * NT;
        w[2 * q] = __uint_as_float(vz[q].y) * 0.f; w[2 * q + 1] = __uint_as_float(vx[q].y) * 0.f;
        if (e < n) {
            const int row = e / NL, col = e % NL, mc = col < ZZC ? col : NL - col;
            bad += (vz[q].x != tag_of(p.epoch, item, 2, 0, (k0 + row) * NL + mc)) + (vx[q].x != tag_of(p.epoch, item, 2, 1, k0 * NL + e));
        }
    }
    const float acc = burn(w, p.f3);
    if (bad) atomicAdd(p.err, bad);
    lds[tid] = acc;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = tid + q * NT;
        if (e < n) xz[k0 * NL + e] = make_float2(__uint_as_float(tag_of(p.epoch, item, 3, 1, k0 * NL + e)), acc);
    }
}

// ---- separate launches ------------------------------------------------------------------------------------------------
// item of a block: blocks of one item on one XCD (block b runs on XCD b % 8), as the library's xcd_coords does
__device__ __forceinline__ void coords(int tiles, int n_items, int& tile, int& item) {
    const int L = blockIdx.x, full = (n_items / 8) * 8;
    if (L < full * tiles) { const int x = L & 7, q = L >> 3; item = (q / tiles) * 8 + x; tile = q % tiles; }
    else { const int r = L - full * tiles; item = full + r / tiles; tile = r % tiles; }
}
template <int PH> __global__ __launch_bounds__(NT) void k_sep(Params p) {
    extern __shared__ float lds[];
    int tile, item;
    coords(PH == 1 ? T1 : PH == 2 ? T2 : PH == 3 ? T3 : T4, p.n_items, tile, item);
    if (PH == 1) phase1(p, item, tile, lds, threadIdx.x);
    else if (PH == 2) phaseA<2>(p, item, tile, lds, threadIdx.x);
    else if (PH == 3) phase3(p, item, tile, lds, threadIdx.x);
    else phaseA<4>(p, item, tile, lds, threadIdx.x);
}

// ---- fused persistent launch --------------------------------------------------------------------------------------------
// Workgroups are phase-specialised: a workgroup that ran all four bodies in one loop would carry the union of their register
// live ranges (146 VGPRs here against 98 for the largest body: three waves per SIMD instead of five), and the real bodies
// also differ in threads and LDS.  The k-th workgroup to arrive on an XCD takes role pattern[k % 16]; every role pulls its
// tiles from its own per-XCD counter in item order.  Dependencies: phase p of item j waits for phase p-1 of item j; phase 1
// of item j waits for phase 4 of item j - W (W items in flight per XCD: what has to stay in the 4 MiB L2).  Every wait is
// for work that workgroups of another role -- always resident -- pull without waiting on this one: no deadlock.
struct Fused {
    unsigned* head;                // [8][4] ticket counters
    unsigned* done;                // [items][4] completed tiles per (item, phase)
    unsigned* census;              // [8] arrivals per XCD, [8..16) polls, [16..24) spare
    int window;                    // W
    unsigned char pattern[16];     // role (0..3) of the k-th arrival, k mod 16
};
__device__ __forceinline__ bool wait_ge(const unsigned* w, unsigned need, unsigned* err, unsigned* polls_out) {
    unsigned polls = 0;
    while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(16); ++polls;
        // bounded spin (~0.2 s), and one time-out ends every other wait at once: report instead of hanging
        if (polls > 300000u || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0x10000u) { atomicAdd(err, 0x10000u); break; }
    }
    if (polls) atomicAdd(polls_out, polls);
    return true;
}
template <int PH>
__device__ __forceinline__ void role_loop(const Params& p, const Fused& f, unsigned xcc, float* lds, unsigned* s_ticket) {
    constexpr int TP = PH == 1 ? T1 : PH == 2 ? T2 : PH == 3 ? T3 : T4;
    constexpr int TPREV = PH == 2 ? T1 : PH == 3 ? T2 : PH == 4 ? T3 : T4;
    const int n_local = (p.n_items - (int)xcc + 7) / 8;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) *s_ticket = atomicAdd(f.head + xcc * 4 + (PH - 1), 1u);
        __syncthreads();
        const unsigned t = *s_ticket;
        if (t >= (unsigned)(n_local * TP)) break;
        const int j = (int)t / TP, tile = (int)t % TP, item = __builtin_amdgcn_readfirstlane((int)xcc + 8 * j);
        if (threadIdx.x == 0) {
            if (PH > 1) wait_ge(f.done + item * 4 + (PH - 2), TPREV, p.err, f.census + 8 + xcc);
            else if (j >= f.window) wait_ge(f.done + (item - 8 * f.window) * 4 + 3, TPREV, p.err, f.census + 8 + xcc);
        }
        __syncthreads();
        // (per-thread index arithmetic is re-derived per tile from an opaque copy of the thread index: hoisted out of the loop
        // it sits in ~45 VGPRs for the loop's whole length -- 141 instead of 98 registers for the phase-3 role)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (PH == 1) phase1(p, item, tile, lds, tid);
        else if (PH == 2) phaseA<2>(p, item, tile, lds, tid);
        else if (PH == 3) phase3(p, item, tile, lds, tid);
        else phaseA<4>(p, item, tile, lds, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave: its stores have reached the XCD's L2
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(f.done + item * 4 + (PH - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
#ifndef PROBE_WPS
#define PROBE_WPS 5
#endif
__global__ __launch_bounds__(NT, PROBE_WPS) void k_fused(const Params* __restrict__ pp, Fused f) {
    extern __shared__ float lds[];
    __shared__ unsigned s_ticket, s_role;
    const Params& p = *pp;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    if (threadIdx.x == 0) s_role = f.pattern[atomicAdd(f.census + xcc, 1u) & 15u];
    __syncthreads();
    const unsigned role = s_role;
    if (role == 0) role_loop<1>(p, f, xcc, lds, &s_ticket);
    else if (role == 1) role_loop<2>(p, f, xcc, lds, &s_ticket);
    else if (role == 2) role_loop<3>(p, f, xcc, lds, &s_ticket);
    else role_loop<4>(p, f, xcc, lds, &s_ticket);
}

// ---- fused, generic workgroups: any workgroup runs any phase ------------------------------------------------------------
// One ordered ticket list per XCD: tiles of (local item i, phase p) are issued in slot D*i + G*(p-1), the groups of a slot
// interleaved tile by tile.  A ticket only ever waits for tickets issued EARLIER in the same list (deadlock-free), and
// (D, G) sets how many items are in flight: (1, 1) four, (2, 1) two, (4, 1) one.
struct Generic {
    const unsigned* list; int list_len[8]; int list_off[8];
    unsigned* head; unsigned* done; unsigned* census;
};
__global__ __launch_bounds__(NT, PROBE_WPS) void k_generic(const Params* __restrict__ pp, Generic f) {
    extern __shared__ float lds[];
    __shared__ unsigned s_ticket;
    const Params& p = *pp;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    if (threadIdx.x == 0) atomicAdd(f.census + xcc, 1u);
    const unsigned* list = f.list + f.list_off[xcc];
    const int len = f.list_len[xcc];
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_ticket = atomicAdd(f.head + xcc * 4, 1u);
        __syncthreads();
        const unsigned t = s_ticket;
        if (t >= (unsigned)len) break;
        const unsigned w = list[t];
        const int tile = w & 255, ph = ((w >> 8) & 3) + 1, item = __builtin_amdgcn_readfirstlane((int)(w >> 10));
        if (ph > 1) {
            if (threadIdx.x == 0) wait_ge(f.done + item * 4 + (ph - 2), ph == 2 ? T1 : ph == 3 ? T2 : T3, p.err, f.census + 8 + xcc);
            __syncthreads();
        }
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (ph == 1) phase1(p, item, tile, lds, tid);
        else if (ph == 2) phaseA<2>(p, item, tile, lds, tid);
        else if (ph == 3) phase3(p, item, tile, lds, tid);
        else phaseA<4>(p, item, tile, lds, tid);
        if (ph < 4) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(f.done + item * 4 + (ph - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static int arg_i(int argc, char** argv, const char* name, int def) {
    for (int i = 1; i + 1 < argc; ++i) if (!strcmp(argv[i], name)) return atoi(argv[i + 1]);
    return def;
}

int main(int argc, char** argv) {
    const int n = arg_i(argc, argv, "--items", 256), reps = arg_i(argc, argv, "--reps", 10);
    const int W = arg_i(argc, argv, "--window", 2), wpc = arg_i(argc, argv, "--wpc", 5);
    const int r1 = arg_i(argc, argv, "--r1", 5), r2 = arg_i(argc, argv, "--r2", 5), r3 = arg_i(argc, argv, "--r3", 4), r4 = arg_i(argc, argv, "--r4", 2);
    const int bypass = arg_i(argc, argv, "--bypass", 1), mode = arg_i(argc, argv, "--mode", 3);      // mode bit 0: separate, bit 1: fused (phase-specialised roles), bit 2: fused (generic workgroups)
    LDS_BYTES = arg_i(argc, argv, "--lds", LDS_BYTES);
    Params p{};
    // defaults: the VALU instructions per complex element of the real kernels (VALU-alone time x 1024 SIMDs / 1.31 ns)
    p.f1 = arg_i(argc, argv, "--f1", 60); p.f2 = arg_i(argc, argv, "--f2", 128); p.f3 = arg_i(argc, argv, "--f3", 87); p.f4 = arg_i(argc, argv, "--f4", 79);
    p.n_items = n; p.l1_bypass = bypass;
    float2 *X, *Z, *S; unsigned* err; float* sink;
    CK(hipMalloc(&X, sizeof(float2) * (size_t)n * PLANE)); CK(hipMalloc(&Z, sizeof(float2) * (size_t)n * PLANE));
    CK(hipMalloc(&S, sizeof(float2) * (size_t)n * 2 * PLANE)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(X, 0, sizeof(float2) * (size_t)n * PLANE)); CK(hipMemset(Z, 0, sizeof(float2) * (size_t)n * PLANE));
    CK(hipMemset(S, 0xFF, sizeof(float2) * (size_t)n * 2 * PLANE)); CK(hipMemset(err, 0, 4));
    p.X = X; p.Z = Z; p.S = S; p.err = err; p.sink = sink;

    // roles of the k-th arrival on an XCD (k mod 16), interleaved; default 5 : 5 : 4 : 2 ~ the phases' shares of the stage's time
    Fused f{};
    f.window = W;
    { const char* pat = "0123012301201201"; int cnt[4] = { r1, r2, r3, r4 };
      if (r1 + r2 + r3 + r4 == 16) {
          int k = 0; while (k < 16) for (int r = 0; r < 4 && k < 16; ++r) if (cnt[r] > 0) { f.pattern[k++] = (unsigned char)r; --cnt[r]; }
      } else for (int k = 0; k < 16; ++k) f.pattern[k] = (unsigned char)(pat[k] - '0'); }
    const int D = arg_i(argc, argv, "--D", 1), G = arg_i(argc, argv, "--G", 1);
    std::vector<unsigned> list; Generic gq{};
    { const int tiles_of[4] = { T1, T2, T3, T4 };
      for (int x = 0; x < 8; ++x) {
        gq.list_off[x] = (int)list.size();
        std::vector<int> items; for (int i = x; i < n; i += 8) items.push_back(i);
        const int nslots = items.empty() ? 0 : D * ((int)items.size() - 1) + G * 3 + 1;
        for (int sl = 0; sl < nslots; ++sl) {
            std::vector<std::pair<int, int>> groups;         // (item, phase-1) issued in this slot, oldest phase first
            for (int ph = 3; ph >= 0; --ph) { const int r = sl - G * ph; if (r >= 0 && r % D == 0 && r / D < (int)items.size()) groups.push_back({ items[r / D], ph }); }
            int mt = 0; for (auto& g : groups) mt = std::max(mt, tiles_of[g.second]);
            for (int t = 0; t < mt; ++t) for (auto& g : groups) if (t < tiles_of[g.second]) list.push_back(((unsigned)g.first << 10) | ((unsigned)g.second << 8) | (unsigned)t);
        }
        gq.list_len[x] = (int)list.size() - gq.list_off[x];
      } }
    unsigned* d_list; CK(hipMalloc(&d_list, 4 * list.size() + 4)); CK(hipMemcpy(d_list, list.data(), 4 * list.size(), hipMemcpyHostToDevice));
    unsigned *d_head, *d_done, *d_census;
    CK(hipMalloc(&d_head, 128)); CK(hipMalloc(&d_done, 16 * (size_t)n)); CK(hipMalloc(&d_census, 96));
    f.head = d_head; f.done = d_done; f.census = d_census;
    gq.list = d_list; gq.head = d_head; gq.done = d_done; gq.census = d_census;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_generic), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));

    Params* d_params; CK(hipMalloc(&d_params, sizeof(Params)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned epoch = 1;
    auto run_sep = [&]() {
        p.epoch = epoch++;
        hipLaunchKernelGGL(k_sep<1>, dim3(T1 * n), dim3(NT), LDS_BYTES, st, p);
        hipLaunchKernelGGL(k_sep<2>, dim3(T2 * n), dim3(NT), LDS_BYTES, st, p);
        hipLaunchKernelGGL(k_sep<3>, dim3(T3 * n), dim3(NT), LDS_BYTES, st, p);
        hipLaunchKernelGGL(k_sep<4>, dim3(T4 * n), dim3(NT), LDS_BYTES, st, p);
    };
    auto run_fused = [&]() {
        p.epoch = epoch++;
        CK(hipMemsetAsync(d_head, 0, 128, st)); CK(hipMemsetAsync(d_done, 0, 16 * (size_t)n, st)); CK(hipMemsetAsync(d_census, 0, 96, st));
        CK(hipMemcpyAsync(d_params, &p, sizeof(Params), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_fused, dim3(256 * wpc), dim3(NT), LDS_BYTES, st, (const Params*)d_params, f);
    };
    auto run_generic = [&]() {
        p.epoch = epoch++;
        CK(hipMemsetAsync(d_head, 0, 128, st)); CK(hipMemsetAsync(d_done, 0, 16 * (size_t)n, st)); CK(hipMemsetAsync(d_census, 0, 96, st));
        CK(hipMemcpyAsync(d_params, &p, sizeof(Params), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_generic, dim3(256 * wpc), dim3(NT), LDS_BYTES, st, (const Params*)d_params, gq);
    };
    auto time_it = [&](auto&& fn, const char* name) {
        fn(); fn(); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) fn();
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h_err; CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
        printf("%-9s %8.4f ms per batch of %d items   mismatching words (all launches so far): %u\n", name, ms / reps, n, h_err);
        return ms / reps;
    };
    printf("rotation-stage probe: %d items, tiles per item %d/%d/%d/%d, dummy FMAs per element %d/%d/%d/%d, consumer loads %s\n",
           n, T1, T2, T3, T4, p.f1, p.f2, p.f3, p.f4, bypass ? "sc1 (L1 bypass)" : "plain");
    float a = 0, b = 0;
    if (mode & 1) a = time_it(run_sep, "separate");
    if (mode & 2) {
        b = time_it(run_fused, "fused");
        unsigned c[24]; CK(hipMemcpy(c, d_census, 96, hipMemcpyDeviceToHost));
        printf("fused: window %d items per XCD, roles %d:%d:%d:%d, %d workgroups per CU; last launch: workgroups per XCD", W, r1, r2, r3, r4, wpc);
        for (int x = 0; x < 8; ++x) printf(" %u", c[x]);
        unsigned long long polls = 0; for (int x = 0; x < 8; ++x) polls += c[8 + x];
        printf("; dependency polls %llu\n", polls);
    }
    if ((mode & 3) == 3) printf("fused / separate = %.3f\n", b / a);
    if (mode & 4) {
        const float c = time_it(run_generic, "generic");
        unsigned cz[24]; CK(hipMemcpy(cz, d_census, 96, hipMemcpyDeviceToHost));
        unsigned long long polls = 0; for (int x = 0; x < 8; ++x) polls += cz[8 + x];
        printf("generic: lag D=%d G=%d (%s items in flight per XCD), %d workgroups per CU; workgroups per XCD", D, G, D == 1 ? "4" : D == 2 ? "2" : "1-2", wpc);
        for (int x = 0; x < 8; ++x) printf(" %u", cz[x]);
        printf("; dependency polls %llu over %zu tickets\n", polls, list.size());
        if (mode & 1) printf("generic / separate = %.3f\n", c / a);
    }
    return 0;
}
