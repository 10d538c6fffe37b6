// kcc_kernels.hip -- gfx950 kernels of the KCC front end (register-resident FFT engine, kcc_fft2.h).
//
// Two kernel families implement every 2-D real FFT of the path (reference correlation_flow.cc:53-77):
//   A-type: 16 lines along the halved axis (rows) per workgroup; real<->half-complex via one complex FFT
//           of length rows/2; the spectrum side is accessed transposed in 128-byte segments, with the
//           r2c split / c2r merge fused into that transposed access.
//   B-type: LK contiguous spectrum lines (length cols) per workgroup, loaded straight into registers.
// Spectra are stored k-major ([rows/2+1][cols], cols contiguous) so the B pass is fully coalesced.
// All pointwise work of the path (|F|, X conj Z, kernel function, ridge solve, max, arg-max, PSR moments,
// polar / rotation gathers) is fused into these passes and runs on registers.
#include "kcc_kernels.h"
#include "kcc_fft2.h"
#include "kcc_pointwise.h"
#include "kcc_tune.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace kcc {

// ------------------------------------------------------------------------------------------------
// instantiated FFT lengths (plans: kcc_fft2.h PlanFor<>)
// ------------------------------------------------------------------------------------------------
#define KCC_HALF_LIST(X) X(30) X(60) X(120) X(224) X(240) X(256) X(360) X(384) X(600)
#define KCC_LINE_LIST(X) X(80) X(160) X(320) X(448) X(480) X(512) X(640) X(752) X(848) X(1024) X(1280) X(1600)

bool fft_half_supported(int h) {
#define X(n) if (h == n) return true;
    KCC_HALF_LIST(X)
#undef X
    return false;
}
bool fft_line_supported(int n_) {
#define X(n) if (n_ == n) return true;
    KCC_LINE_LIST(X)
#undef X
    return false;
}
PlanDesc plan_desc_inv(int n_) {
    PlanDesc d{ n_, 0, { 1, 1, 1 }, 0, 0 };
#define X(n) if (n_ == n) { using P = PlanInv<n>; d.np = P::NP; d.r[0] = P::R1; d.r[1] = P::R2; d.r[2] = P::R3; d.t = P::T; d.prime = P::PRIME ? P::R2 : 0; }
    KCC_HALF_LIST(X)
#undef X
    return d;
}
PlanDesc plan_desc_alt(int n_) {
    PlanDesc d{ n_, 0, { 1, 1, 1 }, 0, 0 };
#define X(n) if (n_ == n) { using P = PlanAlt<n>; d.np = P::NP; d.r[0] = P::R1; d.r[1] = P::R2; d.r[2] = P::R3; d.t = P::T; d.prime = P::PRIME ? P::R2 : 0; }
    KCC_LINE_LIST(X)
#undef X
    return d;
}
PlanDesc plan_desc(int n_) {
    PlanDesc d{ n_, 0, { 1, 1, 1 }, 0, 0 };
#define X(n) if (n_ == n) { using P = PlanFor<n>; d.np = P::NP; d.r[0] = P::R1; d.r[1] = P::R2; d.r[2] = P::R3; d.t = P::T; d.prime = P::PRIME ? P::R2 : 0; }
    KCC_HALF_LIST(X)
    KCC_LINE_LIST(X)
#undef X
    return d;
}

#ifndef KCC_GATHER_GROUP
#define KCC_GATHER_GROUP 4
#endif
#ifndef KCC_WAVE_LOCAL
#define KCC_WAVE_LOCAL 1
#endif
#ifndef KCC_ALX
#define KCC_ALX 16
#endif
#ifndef KCC_WAVE_REGION
#define KCC_WAVE_REGION 1
#endif
// Performance ablation (tuning builds only: -DKCC_ABLATE, tools/ablate.sh; results become garbage).  $NIK_ABLATE bits:
// 1 no loads / gathers, 2 no stores, 4 no FFT, 8 exit at once (dispatch cost only), 16 no gather staging, 32 no gather sampling,
// 64 no u8 frame-store copy.  The release library contains none of it (ABL() is a compile-time false).
#ifdef KCC_ABLATE
static int ablate_flags() { static const int f = tune_env("NIK_ABLATE") ? atoi(tune_env("NIK_ABLATE")) : 0; return f; }
#define ABL(a, bit) (((a).ablate & (bit)) != 0)
#else
static int ablate_flags() { return 0; }
#define ABL(a, bit) false
#endif

// Upper bounds by deletion (tuning builds only; results become garbage, TIMES stay meaningful): what a remedy could return at
// most, measured before it is built (round 6).
//   KCC_UB_NOMOMENTS  the arg-max kernels accumulate no PSR moments   -> ceiling of "moments by Parseval in solve_inv"
//   KCC_UB_NOZZFWD    solve_inv skips the forward transform of Kzz    -> twice the ceiling of "Kzz real-line pairs" (5 transforms per line pair instead of 6)
//   KCC_UB_POLAR8     the polar gather reads 8-byte sample entries    -> ceiling of "8-byte polar sample table"
// (-DKCC_UB_TIMING_ONLY: the same deletion on the RELEASE code generation -- the ablation build's ABL() branches change register
// allocation, e.g. its polar kernel takes 0.255 ms against the release's 0.226 -- for a variant library that is only ever timed)
#if !defined(KCC_ABLATE) && !defined(KCC_UB_TIMING_ONLY) && (defined(KCC_UB_NOMOMENTS) || defined(KCC_UB_NOZZFWD) || defined(KCC_UB_POLAR8))
#error "KCC_UB_* variants produce wrong results: tuning library (-DKCC_ABLATE) only"
#endif

// lines per A-type workgroup: 16 cf2 = one 128-byte segment per spectrum row; the long polar lines (h = 360)
// use 8 so that twice as many independent workgroups fit in a CU's LDS (their phases overlap better)
#ifndef KCC_ALX360
#define KCC_ALX360 8
#endif
__host__ __device__ constexpr int a_lx(int hh) { return hh >= 360 ? KCC_ALX360 : KCC_ALX; }

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// (wrap_idx, bilerp, affine terms, kernel_value: kcc_pointwise.h -- shared with the generic-size kernels, kcc_generic.hip)

// Workgroups of one item share its source plane (gathers): put them on the same XCD (the dispatcher places
// linear block b on XCD b % 8), so the plane is fetched into one L2 instead of eight.  Speed only.
// rev: walk the items back to front.  Consecutive kernels of a lane alternate the direction (set_launch_reverse), so a
// consumer starts with the items its producer wrote LAST -- the ones still in the 256 MiB Infinity Cache
// (tools/probes/mall_order_probe.hip: +22 % on a copy chain whose planes are about the cache's size).  Speed only.
__device__ __forceinline__ void xcd_coords_of(int L, int nbx, int n_items, int& bx, int& item, int rev = 0);
__device__ __forceinline__ void xcd_coords(int nbx, int n_items, int& bx, int& item, int rev = 0) {
    xcd_coords_of((int)blockIdx.x, nbx, n_items, bx, item, rev);
}
__device__ __forceinline__ void xcd_coords_of(int L, int nbx, int n_items, int& bx, int& item, int rev) {
    const int full_items = (n_items / 8) * 8;
    if (L < full_items * nbx) {
        const int xcd = L & 7, q = L >> 3;
        item = (q / nbx) * 8 + xcd;
        bx = q % nbx;
    } else {
        const int r = L - full_items * nbx;
        item = full_items + r / nbx;
        bx = r % nbx;
    }
    if (rev) item = n_items - 1 - item;
}
static thread_local int g_launch_rev = 0;
void set_launch_reverse(int rev) { g_launch_rev = rev ? 1 : 0; }

// ------------------------------------------------------------------------------------------------
// u8 row-major -> f32 column-major, /255  (utils.cc:110-118)
// ------------------------------------------------------------------------------------------------
// u8 frame-store image -> f32 plane of the same slot (frames that arrived as u8 and are needed as f32: nik_frame_export,
// batches mixing u8 and f32 frames).  64 x 64 tile per workgroup: rows are read as uchar4 (64-byte row segments),
// transposed through LDS and written as float4 along y (256-byte column segments).  Requires W % 4 == 0 and H % 4 == 0.
// Image planes are stored with column pitch PH >= H + 4: rows H..H+3 of a column repeat its rows 0..3, so a vertical
// BORDER_WRAP tap pair (H-1, 0) is one contiguous 8-byte load (rot_sample).
__global__ __launch_bounds__(256) void k_cvt_u8(const uint8_t* __restrict__ src, size_t src_stride, int src_pitch, const int* __restrict__ slot,
                                                float* __restrict__ arena, int H, int W, int PH) {
    __shared__ float tile[64][65];                          // [x][y], odd pitch
    const int item = blockIdx.z, tid = threadIdx.x;
    const uint8_t* in = src + (size_t)slot[item] * src_stride;
    float* out = arena + (size_t)slot[item] * PH * W;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = r0 + (tid >> 4) + 16 * it, c = c0 + 4 * (tid & 15);
        if (r < H && c < W) {
            const uchar4 v = *reinterpret_cast<const uchar4*>(in + (size_t)r * src_pitch + c);
            const int y = (tid >> 4) + 16 * it, x = 4 * (tid & 15);
            tile[x + 0][y] = (float)v.x / 255.0f; tile[x + 1][y] = (float)v.y / 255.0f;
            tile[x + 2][y] = (float)v.z / 255.0f; tile[x + 3][y] = (float)v.w / 255.0f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int x = (tid >> 4) + 16 * it, y = 4 * (tid & 15);
        const int c = c0 + x, r = r0 + y;
        if (c < W && r < H) {
            const float4 v = make_float4(tile[x][y], tile[x][y + 1], tile[x][y + 2], tile[x][y + 3]);
            *reinterpret_cast<float4*>(out + (size_t)c * PH + r) = v;
            if (r == 0) *reinterpret_cast<float4*>(out + (size_t)c * PH + H) = v;      // wrap rows
        }
    }
}

// cv::cvtColor(COLOR_RGB2GRAY / COLOR_BGR2GRAY) on 8-bit data: (R*4899 + G*9617 + B*1868 + 8192) >> 14  [recalled]
__device__ __forceinline__ unsigned luma8(unsigned r, unsigned g, unsigned b) { return (r * 4899u + g * 9617u + b * 1868u + 8192u) >> 14; }
// 16 pixels per thread: three 16-byte loads (48 bytes of interleaved RGB), one 16-byte store -- the scalar form (one pixel
// and three byte loads per thread) ran at 1.6 TB/s and was 8.8 % of every configs[3] step
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_rgb2gray16(const u32x4* __restrict__ rgb, u32x4* __restrict__ gray, size_t ngroups, int bgr) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ngroups) return;
    const u32x4 a = __builtin_nontemporal_load(rgb + 3 * i), b = __builtin_nontemporal_load(rgb + 3 * i + 1), c = __builtin_nontemporal_load(rgb + 3 * i + 2);
    const unsigned w[12] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w };
    unsigned o[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int k = 3 * p;
        const unsigned c0 = (w[k >> 2] >> (8 * (k & 3))) & 255u;
        const unsigned c1 = (w[(k + 1) >> 2] >> (8 * ((k + 1) & 3))) & 255u;
        const unsigned c2 = (w[(k + 2) >> 2] >> (8 * ((k + 2) & 3))) & 255u;
        o[p >> 2] |= luma8(bgr ? c2 : c0, c1, bgr ? c0 : c2) << (8 * (p & 3));
    }
    u32x4 r; r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
    __builtin_nontemporal_store(r, gray + i);
}
// the pixels behind the last whole group of 16 (and unaligned buffers): one pixel per thread
__global__ void k_rgb2gray(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ gray, size_t first, size_t npix, int bgr) {
    const size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const uint8_t* p = rgb + 3 * i;
    gray[i] = (uint8_t)luma8(bgr ? p[2] : p[0], p[1], bgr ? p[0] : p[2]);
}
// 2 x 2 box filter, rounded: one pyramid level (u8 row-major H x W -> H/2 x W/2); no reference counterpart
__global__ void k_downsample_u8(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W) {
    const int Wo = W / 2, Ho = H / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x, item = blockIdx.y;
    if (i >= Ho * Wo) return;
    const int r = i / Wo, c = i - r * Wo;
    const uint8_t* p = in + (size_t)item * H * W + (size_t)(2 * r) * W + 2 * c;
    out[(size_t)item * Ho * Wo + i] = (uint8_t)((p[0] + p[1] + p[W] + p[W + 1] + 2) >> 2);
}
void launch_downsample_u8(hipStream_t s, int n, const uint8_t* in, uint8_t* out, int H, int W) {
    hipLaunchKernelGGL(k_downsample_u8, dim3(((H / 2) * (W / 2) + 255) / 256, n), dim3(256), 0, s, in, out, H, W);
}
// D consecutive pyramid levels in one launch: a thread owns a 2^D x 2^D block of the source, reads it with 8-byte loads and
// writes its share of every level (each level is the rounded 2 x 2 box filter of the level above it -- the same integers the
// chained k_downsample_u8 calls produce).  Frames [0, na) come from `a`, frames [na, nf) from `b` (the key and the current
// frames of a batch arrive in two buffers); level d's frames are stored back to back in out[d - 1].
template <int D>
__global__ __launch_bounds__(256) void k_downsample_pyr(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int na, int H, int W,
                                                        uint8_t* __restrict__ o1, uint8_t* __restrict__ o2, uint8_t* __restrict__ o3) {
    constexpr int S = 1 << D;
    const int Wb = W / S, Hb = H / S;
    const int i = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    if (i >= Hb * Wb) return;
    const int r = i / Wb, c = i - r * Wb;
    const uint8_t* src = (f < na ? a + (size_t)f * H * W : b + (size_t)(f - na) * H * W) + (size_t)(r * S) * W + c * S;
    unsigned v[S][S];
#pragma unroll
    for (int y = 0; y < S; ++y) {
        if (S == 8) {
            const uint2 w = *reinterpret_cast<const uint2*>(src + (size_t)y * W);
#pragma unroll
            for (int x = 0; x < 4; ++x) { v[y][x] = (w.x >> (8 * x)) & 255u; v[y][4 + x] = (w.y >> (8 * x)) & 255u; }
        } else if (S == 4) {
            const unsigned w = *reinterpret_cast<const unsigned*>(src + (size_t)y * W);
#pragma unroll
            for (int x = 0; x < 4; ++x) v[y][x] = (w >> (8 * x)) & 255u;
        } else {
            const unsigned short w = *reinterpret_cast<const unsigned short*>(src + (size_t)y * W);
            v[y][0] = w & 255u; v[y][1] = w >> 8;
        }
    }
    uint8_t* const outs[3] = { o1, o2, o3 };
#pragma unroll
    for (int d = 1; d <= D; ++d) {
        const int n = S >> d;                                  // this thread's n x n pixels of level d
#pragma unroll
        for (int y = 0; y < n; ++y)
#pragma unroll
            for (int x = 0; x < n; ++x) v[y][x] = (v[2 * y][2 * x] + v[2 * y][2 * x + 1] + v[2 * y + 1][2 * x] + v[2 * y + 1][2 * x + 1] + 2u) >> 2;
        const int Wd = W >> d, Hd = H >> d;
        uint8_t* o = outs[d - 1] + (size_t)f * Hd * Wd + (size_t)(r * n) * Wd + c * n;
#pragma unroll
        for (int y = 0; y < n; ++y) {
            if (n == 4) *reinterpret_cast<unsigned*>(o + (size_t)y * Wd) = v[y][0] | (v[y][1] << 8) | (v[y][2] << 16) | (v[y][3] << 24);
            else if (n == 2) *reinterpret_cast<unsigned short*>(o + (size_t)y * Wd) = (unsigned short)(v[y][0] | (v[y][1] << 8));
            else o[(size_t)y * Wd] = (uint8_t)v[y][0];
        }
    }
}
// steps in [1, 3]; H and W divisible by 2^steps and every pointer aligned to 2^steps bytes (the caller checks)
void launch_downsample_pyr(hipStream_t s, int steps, const uint8_t* a, const uint8_t* b, int na, int nf, int H, int W, uint8_t* const* out) {
    const int S = 1 << steps;
    dim3 grid(((H / S) * (W / S) + 255) / 256, nf), block(256);
    if (steps == 1) hipLaunchKernelGGL(k_downsample_pyr<1>, grid, block, 0, s, a, b, na, H, W, out[0], nullptr, nullptr);
    else if (steps == 2) hipLaunchKernelGGL(k_downsample_pyr<2>, grid, block, 0, s, a, b, na, H, W, out[0], out[1], nullptr);
    else hipLaunchKernelGGL(k_downsample_pyr<3>, grid, block, 0, s, a, b, na, H, W, out[0], out[1], out[2]);
}
void launch_rgb2gray(hipStream_t s, const uint8_t* rgb, uint8_t* gray, size_t npix, int bgr) {
    const bool aligned = (reinterpret_cast<uintptr_t>(rgb) % 16 == 0) && (reinterpret_cast<uintptr_t>(gray) % 16 == 0);
    const size_t groups = aligned ? npix / 16 : 0, done = groups * 16;
    if (groups) hipLaunchKernelGGL(k_rgb2gray16, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const u32x4*>(rgb),
                                   reinterpret_cast<u32x4*>(gray), groups, bgr);
    if (done < npix) hipLaunchKernelGGL(k_rgb2gray, dim3((unsigned)((npix - done + 255) / 256)), dim3(256), 0, s, rgb, gray, done, npix, bgr);
}

void launch_cvt_u8(hipStream_t s, int n, const uint8_t* arena_u8, size_t u8_stride, int u8_pitch, const int* d_slot, float* arena_img,
                   int H, int W, int PH) {
    dim3 grid((W + 63) / 64, (H + 63) / 64, n), block(256);
    hipLaunchKernelGGL(k_cvt_u8, grid, block, 0, s, arena_u8, u8_stride, u8_pitch, d_slot, arena_img, H, W, PH);
}
// rows 0..3 of every column copied behind row H-1 (after a host import of a plane)
__global__ void k_img_wrap(float* __restrict__ img, int H, int W, int PH) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < W) *reinterpret_cast<float4*>(img + (size_t)c * PH + H) = *reinterpret_cast<const float4*>(img + (size_t)c * PH);
}
void launch_img_wrap(hipStream_t s, float* img, int H, int W, int PH) {
    hipLaunchKernelGGL(k_img_wrap, dim3((W + 255) / 256), dim3(256), 0, s, img, H, W, PH);
}

// ------------------------------------------------------------------------------------------------
// Camera::UndistortImage (camera.cc:92-93): cv::remap(8UC1, CV_16SC2 + CV_16UC1 maps, INTER_LINEAR, BORDER_CONSTANT 0)
// ------------------------------------------------------------------------------------------------
// remapBilinear<FixedPtCast<int, uchar, 15>>: 15-bit integer weights -- exact for 1/32 px fractions:
// (32-fx)(32-fy)*32 ... , sum 32768 -- and out-of-image taps = border value 0.  Returns the u8 result.
__device__ __forceinline__ int undistort_px(const uint8_t* __restrict__ in, int H, int W, int2 m1, int m2) {
    const int sx = m1.x, sy = m1.y, fx = m2 & 31, fy = (m2 >> 5) & 31;
    const bool x0 = (unsigned)sx < (unsigned)W, x1 = (unsigned)(sx + 1) < (unsigned)W;
    const bool y0 = (unsigned)sy < (unsigned)H, y1 = (unsigned)(sy + 1) < (unsigned)H;
    const uint8_t* p = in + (long)sy * W + sx;
    const int t00 = (x0 && y0) ? p[0] : 0, t01 = (x1 && y0) ? p[1] : 0;
    const int t10 = (x0 && y1) ? p[W] : 0, t11 = (x1 && y1) ? p[W + 1] : 0;
    const int acc = ((32 - fx) * (32 - fy) * t00 + fx * (32 - fy) * t01 + (32 - fx) * fy * t10 + fx * fy * t11) * 32;
    return (acc + (1 << 14)) >> 15;
}
// u8 -> u8 (the undistorted image itself, e.g. for the map stitcher)
__global__ __launch_bounds__(256) void k_undistort_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                      const short2* __restrict__ map1, const uint16_t* __restrict__ map2, int H, int W) {
    const int i = blockIdx.x * 256 + threadIdx.x, item = blockIdx.y;
    if (i >= H * W) return;
    const short2 m = map1[i];
    dst[(size_t)item * H * W + i] = (uint8_t)undistort_px(src + (size_t)item * H * W, H, W, make_int2(m.x, m.y), map2[i]);
}
void launch_undistort_u8(hipStream_t s, int n, const uint8_t* d_in, uint8_t* d_out, const int16_t* map1, const uint16_t* map2, int H, int W) {
    hipLaunchKernelGGL(k_undistort_u8, dim3((H * W + 255) / 256, n), dim3(256), 0, s, d_in, d_out,
                       reinterpret_cast<const short2*>(map1), map2, H, W);
}

// ------------------------------------------------------------------------------------------------
// MapStitcher::AddImageToOccupancy (map_stitcher.cc:36-133): scatter of one key frame into the occupancy cells
// ------------------------------------------------------------------------------------------------
// MapStitcher::ComputeCellPosition (map_stitcher.cc:24-34): floor division into (cell, position in cell)
__device__ __forceinline__ void cell_position(int x, int size, int& cell, int& pos) {
    cell = x >= 0 ? x / size : (x - size + 1) / size;
    pos = x - cell * size;
}
// one thread per source pixel (i = column, j = row): map position = trunc(R * (i - cx, j - cy) + t), evaluated in
// double exactly as the reference's Eigen expressions ((R00 * (i - cx) + X) + R01 * (j - cy), no contraction);
// data += 100/255-scaled pixel, weight += 1 in the frame's temporary cells (tmp_data / tmp_weight of cell k at k * size^2)
__global__ void k_stitch_scatter(const uint8_t* __restrict__ img, int H, int W, StitchPose P, int size,
                                 int cx0, int cy0, int ncx, int ncy, int* __restrict__ tmp_data, int* __restrict__ tmp_weight) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i >= W) return;
    const double wi = (double)i - P.cx, hj = (double)j - P.cy;
    const int x = (int)((P.r00 * wi + P.x) + P.r01 * hj);
    const int y = (int)((P.r10 * wi + P.y) + P.r11 * hj);
    int gx, px, gy, py;
    cell_position(x, size, gx, px); cell_position(y, size, gy, py);
    const int k = (gx - cx0) * ncy + (gy - cy0);                   // the frame's cell grid is built from its corner pixels
    if (gx < cx0 || gx >= cx0 + ncx || gy < cy0 || gy >= cy0 + ncy) return;   // (cannot happen for a rotated rectangle)
    const size_t e = (size_t)k * size * size + (size_t)py * size + px;
    // cv::Mat(u8) * (100.0 / 255.0): saturate_cast<uchar>(v * scale), round to nearest (no exact halves exist)
    const int v = (int)__float2int_rn((float)((double)img[(size_t)j * W + i] * (100.0 / 255.0)));
    atomicAdd(tmp_data + e, v);
    atomicAdd(tmp_weight + e, 1);
}
// merge of one temporary cell into the map cell (map_stitcher.cc:113-131), literally:
//   existing cell: data = data * weight + tmp.data * tmp.weight; weight += tmp.weight; data /= weight where weight >= 1
//   new cell:      data = tmp.data; weight = tmp.weight
__global__ void k_stitch_merge(int* __restrict__ data, int* __restrict__ weight, const int* __restrict__ tmp_data,
                               const int* __restrict__ tmp_weight, int n, int existing) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (existing) {
        const int d = data[e] * weight[e] + tmp_data[e] * tmp_weight[e];
        const int w = weight[e] + tmp_weight[e];
        weight[e] = w;
        data[e] = w < 1 ? d : d / w;
    } else {
        data[e] = tmp_data[e]; weight[e] = tmp_weight[e];
    }
}
// flags[k] = 1 iff temporary cell k received any pixel
__global__ void k_stitch_touched(const int* __restrict__ tmp_weight, int csz, int* __restrict__ flags) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (e < csz && tmp_weight[(size_t)k * csz + e] != 0) flags[k] = 1;
}
void launch_stitch_touched(hipStream_t s, const int* tmp_weight, int n_cells, int csz, int* flags) {
    hipLaunchKernelGGL(k_stitch_touched, dim3((csz + 255) / 256, n_cells), dim3(256), 0, s, tmp_weight, csz, flags);
}
void launch_stitch_scatter(hipStream_t s, const uint8_t* img, int H, int W, const StitchPose& P, int size, int cx0, int cy0,
                           int ncx, int ncy, int* tmp_data, int* tmp_weight) {
    hipLaunchKernelGGL(k_stitch_scatter, dim3((W + 255) / 256, H), dim3(256), 0, s, img, H, W, P, size, cx0, cy0, ncx, ncy, tmp_data, tmp_weight);
}
void launch_stitch_merge(hipStream_t s, int* data, int* weight, const int* tmp_data, const int* tmp_weight, int n, int existing) {
    hipLaunchKernelGGL(k_stitch_merge, dim3((n + 255) / 256), dim3(256), 0, s, data, weight, tmp_data, tmp_weight, n, existing);
}

// ------------------------------------------------------------------------------------------------
// A-type kernels
// ------------------------------------------------------------------------------------------------
enum { SRC_PLANE = 0, SRC_ROT = 1,          // f32 column-major planes (nik_intermedium_f32 / nik_frame_import frames)
       SRC_U8 = 3,                          // u8 row-major image tile, /255 on the fly (ConvertMatToNormalizedArray fused)
       SRC_ROT8 = 4,                        // RotateArray from the u8 frame store, source bands staged in LDS
       SRC_POLAR_H = 5, SRC_POLAR_Q = 6, SRC_POLAR_T = 7 };  // polar gather from LDS-staged annulus segments holding rf/2 / 1 /
                                                             // polar_qs_mid(rf) of a thread's first-pass points
enum { EPI_REAL = 0, EPI_KFWD_POLY3 = 1, EPI_ARGMAX = 2, EPI_KFWD_POLYN = 3, EPI_KFWD_GAUSS = 4, EPI_SHIFTED = 5,
       EPI_ARGMAX_WIN = 6 };   // arg-max restricted to a cyclic window per item (coarse-to-fine registration)
__host__ __device__ constexpr bool epi_is_kfwd(int e) { return e == EPI_KFWD_POLY3 || e == EPI_KFWD_POLYN || e == EPI_KFWD_GAUSS; }

// kernel-argument pointers to complex data: the C++ interface of this unit (kcc_kernels.h) speaks HIP's float2, the kernels
// the register-pair type cf2 (kcc_fft.h) -- same memory layout; these wrappers convert at the launch boundary
struct CP {
    cf2* p;
    __host__ __device__ CP() : p(nullptr) {}
    __host__ __device__ CP(float2* q) : p(reinterpret_cast<cf2*>(q)) {}
    __host__ __device__ operator cf2*() const { return p; }
};
struct CCP {
    const cf2* p;
    __host__ __device__ CCP() : p(nullptr) {}
    __host__ __device__ CCP(const float2* q) : p(reinterpret_cast<const cf2*>(q)) {}
    __host__ __device__ operator const cf2*() const { return p; }
};
struct AArgs {
    int rows, cols, hr, n_items, ablate, rev;
    CCP tw_f, tw_i, tw_full;
    CCP twI_f, twI_i;                // tables of PlanInv (spectrum-in kernels)
    // forward source
    const float* src; size_t src_stride; const int* src_idx; int src_pitch;   // image planes: column pitch (>= rows, wrap rows behind)
    const int* rot_tab; const int* rot_index;              // per-angle int tables [adelta W | bdelta W | X0 H | Y0 H]
    int H, W, SP;                                            // polar source: shifted planes, column pitch SP
    int polar_group;                                         // polar tiles per group (divides the tile count): see the kernel's block order
    const uint32_t* polar_chunks; const int* polar_seg_first; const uint4* polar_pts;   // staging descriptors / per-thread sample entries
    // u8 sources: SRC_U8 reads the batch input (row pitch src8_pitch) and copies its tile into the u8 frame store;
    // SRC_ROT8 reads the u8 frame store (row pitch src8_pitch = W + 16, columns 0..15 repeated behind column W-1)
    const uint8_t* src8; size_t src8_stride; int src8_pitch;
    uint8_t* dst8; size_t dst8_stride; int dst8_pitch; const int* dst8_slot;
    float* dbg_plane;                                        // debug tap: the gathered real plane of item 0 (column-major rows x cols)
    // spectrum side
    CP spec; size_t spec_stride; size_t plane_stride;
    int plane_first, n_planes;                               // kernel_fwd: planes [plane_first, plane_first+n_planes) of each item
    int zz_tiles;                                            // kernel_fwd over (zz, xz): > 0 = only this many column tiles of the zz plane (Hermitian half)
    // inverse outputs
    float* real_out; size_t real_stride;
    Partial* partials; int partial_stride;
    const int* win_row; const int* win_col; int win_radius, win_mirror;   // ARGMAX_WIN: per-item window centre; mirror: also centre + rows/2
    int fix_zero;                                            // SHIFTED (mirrored half-plane form): apply RemoveZeroComponent on the way
    KernelFn fn; unsigned* maxbuf; const float* energy;      // maxbuf: per-wave running-max parts [item][2][KCC_MAXPARTS] (float bits)
};

// most waves per SIMD ever asked of the register allocator: the 360-point kernels need ~100 VGPRs (a 96-register cap spilled)
#ifndef KCC_WPS_MAX
#define KCC_WPS_MAX(hh) ((hh) >= 360 ? 4 : 8)
#endif
template <int HH, int LXV, bool INVPLAN> struct ACfg {
    static constexpr int HALF = HH;
    using P = typename std::conditional<INVPLAN, PlanInv<HH>, PlanFor<HH>>::type;
    static constexpr int T = P::T;                       // threads per line
    static constexpr int LX = LXV;
    static constexpr int NT = LX * T;
    // exchange pitch: >= EXT and == T (mod 32) so the lines sharing a wave-instruction use disjoint banks
    static constexpr int EPITCH = ((P::EXT + 31 - (T % 32)) / 32) * 32 + (T % 32);
    static constexpr int NPITCH = HH + 1;                // natural-order pitch: odd -> conflict-free transposes
    static constexpr int LDS_ELEMS = LX * (EPITCH > NPITCH ? EPITCH : NPITCH);
    static constexpr size_t BYTES = (size_t)LDS_ELEMS * sizeof(cf2);
    // Wave-owned LDS regions (wave-local plans, T | 64): wave w holds lines [w LPW, (w+1) LPW).  Their natural-order rows
    // occupy [w LPW NPITCH, (w+1) LPW NPITCH); their exchange buffers are placed INSIDE that range (pitch EPITCH <= NPITCH
    // from the region's start), so between the workgroup-wide load and the workgroup-wide store a wave only ever touches LDS
    // that no other wave touches: the natural -> exchange -> natural hand-overs need a wave-level fence, not an s_barrier,
    // and the waves of a workgroup drift apart instead of marching in lock-step (2 barriers per tile instead of 5 to 7).
    static constexpr int LPW = (64 % T == 0) ? 64 / T : 0;
    static constexpr bool WREG = KCC_WAVE_REGION && KCC_WAVE_LOCAL && LPW > 0 && (LX % (LPW > 0 ? LPW : 1) == 0) && EPITCH <= NPITCH;
    __device__ static __forceinline__ cf2* ex_of(cf2* lds, int line) {
        if constexpr (WREG) return lds + (line / LPW) * (LPW * NPITCH) + (line % LPW) * EPITCH;
        else return lds + line * EPITCH;
    }
    // waves per SIMD the LDS footprint allows: ask the register allocator to fit that occupancy
    static constexpr int BLOCKS = (int)(160 * 1024 / BYTES) > 8 ? 8 : (int)(160 * 1024 / BYTES);
    static constexpr int WPS_ = (BLOCKS * ((NT + 63) / 64) + 3) / 4;
    static constexpr int WPS = WPS_ > KCC_WPS_MAX(HH) ? KCC_WPS_MAX(HH) : (WPS_ < 1 ? 1 : WPS_);
};

// hand-over between a wave's own natural-order rows and its own exchange buffers (ACfg::WREG), else a workgroup barrier
template <bool WREG> __device__ __forceinline__ void region_sync() { line_sync<WREG>(); }
template <int RR>
__device__ __forceinline__ void zero_fill(cf2 (&v)[RR]) {
#pragma unroll
    for (int q = 0; q < RR; ++q) v[q] = mk2(0.f, 0.f);
}
// 8-byte load from a 4-byte aligned address (two vertically adjacent taps)
__device__ __forceinline__ cf2 load2(const float* p) {
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    const f2u v = *reinterpret_cast<const f2u*>(p);
    return mk2(v.x, v.y);
}
// RotateArray (utils.cc:154-161): cv::warpAffine(INTER_LINEAR, BORDER_WRAP).  The fixed-point coordinate
// terms of OpenCV's WarpAffineInvoker (adelta[c], bdelta[c], X0[r], Y0[r]) are tabulated per candidate angle on
// the host, so the device does integer adds only.  nik_create() restricts the aspect ratio so that a rotation
// about the centre keeps every source coordinate within one period: BORDER_WRAP is a single conditional add
// (identical to cv::borderInterpolate there), and saturate_cast<short> can never clip.  Returns dst pixel (r, c).
__device__ __forceinline__ float rot_sample(const float* __restrict__ img, int H, int PH, int W, int ad, int bd, int X0, int Y0) {
    const int X = (X0 + ad) >> 5, Y = (Y0 + bd) >> 5;
    int xa = X >> 5, ya = Y >> 5;
    xa += (xa < 0) ? W : 0; xa -= (xa >= W) ? W : 0;
    ya += (ya < 0) ? H : 0; ya -= (ya >= H) ? H : 0;
    const int xb = (xa + 1 == W) ? 0 : xa + 1;
    // The two taps of a column are adjacent in memory (rows are contiguous) and every column carries a copy of its
    // row 0 behind row H-1 (column pitch PH > H), so the vertical wrap needs no fix-up: one 8-byte load per column.
    // Unsigned 32-bit element offsets (masked to 28 bits, a no-op for any real plane, so the byte offset provably
    // fits 32 bits) from the wave-uniform image base: scalar-base + vector-offset loads, no per-lane 64-bit pointers.
    const unsigned oa = (unsigned)(xa * PH + ya) & 0x0FFFFFFFu, ob = (unsigned)(xb * PH + ya) & 0x0FFFFFFFu;
    const cf2 va = load2(img + oa), vb = load2(img + ob);
    return bilerp(va.x, vb.x, va.y, vb.y, X & 31, Y & 31);
}
#ifndef KCC_FLX360
#define KCC_FLX360 16
#endif
template <int HH> using FCfg = ACfg<HH, (HH >= 360 ? KCC_FLX360 : KCC_ALX), false>;            // forward (real -> spectrum) kernels
// inverse (spectrum -> ...) kernels; the read-only arg-max kernels pick their own tile width
#ifndef KCC_ALX_AM
#define KCC_ALX_AM KCC_ALX
#endif
#ifndef KCC_ALX_AM360
#define KCC_ALX_AM360 KCC_ALX360
#endif
__host__ __device__ constexpr int inv_lx(int hh, int epi) {
    return (epi == EPI_ARGMAX || epi == EPI_ARGMAX_WIN) ? (hh >= 360 ? KCC_ALX_AM360 : KCC_ALX_AM) : a_lx(hh);
}
// Preferred lines per workgroup of the inverse kernels where it divides the plane's columns (run-time choice, LXO template
// argument): 12 lines of the 360-point family = 240 threads = 3.75 waves and 96-byte row segments, against 8 lines = 160
// threads = 2.5 waves (a half-empty third wave) and 64-byte segments (kernel_fwd at 720x480: 0.314 -> 0.291 ms)
#ifndef KCC_ALXP360
#define KCC_ALXP360 12
#endif
#ifndef KCC_ALXP240
#define KCC_ALXP240 0
#endif
__host__ __device__ constexpr int pref_lx(int hh) { return hh == 360 ? KCC_ALXP360 : (hh == 240 ? KCC_ALXP240 : 0); }
inline int inv_lx_for(int hh, int cols, int epi) { const int p = pref_lx(hh); return (p > 0 && cols % p == 0) ? p : inv_lx(hh, epi); }
template <int HH, int EPI, int LXO = 0> using ICfg = ACfg<HH, (LXO > 0 ? LXO : inv_lx(HH, EPI)), true>;

// natural-order packed-FFT lines in LDS -> r2c split -> transposed global store (k-major spectrum).
// All LDS / twiddle reads of a thread are issued before the arithmetic (memory-level parallelism).
// r2c split of one (k, h-k) pair of one line: Z -> X[k], X[h-k]
__device__ __forceinline__ void r2c_pair(cf2 za, cf2 zb, cf2 w, cf2& xk, cf2& xh) {
    const cf2 e = scale(cadd_conj(za, zb), 0.5f);                             // (za + conj zb) / 2
    const cf2 p = cmul(w, scale(csub_conj(za, zb), 0.5f));                    // w^k d,   d = (za - conj zb) / 2
    xk = sub_ib(e, p);                                                        // e - i w d
    xh = conj_add_ib(e, p);                                                   // conj(e + i w d)
}
// c2r merge of one (k, h-k) pair of one line: X[k], X[h-k] -> Z'[k], Z'[h-k]
__device__ __forceinline__ void c2r_pair(cf2 xa, cf2 xb, cf2 w, cf2& zk, cf2& zh) {
    const cf2 sm = cadd_conj(xa, xb), u = cmulc(csub_conj(xa, xb), w);        // xa + conj xb;  conj(w^k) (xa - conj xb)
    zk = add_ib(sm, u);                                                       // sm + i u
    zh = conj_sub_ib(sm, u);                                                  // conj(sm - i u)
}

// natural-order packed-FFT lines in LDS -> r2c split -> transposed global store (k-major spectrum).
// A thread owns the (k, h-k) pair of TWO adjacent lines: every global access is 16 bytes per lane (8 lanes cover a
// 128-byte row segment), and all LDS / twiddle reads are issued before the arithmetic.
template <class C>
__device__ __forceinline__ void a_post_store(const cf2* nat, const cf2* __restrict__ tw_full,
                                             cf2* __restrict__ spec, int cols, int x0, int tid) {
    constexpr int HH = C::HALF, NPITCH = C::NPITCH, NT = C::NT, NP = HH / 2 + 1;
    constexpr int LX2 = C::LX / 2;
    constexpr int TOT = LX2 * NP, ITERS = (TOT + NT - 1) / NT;
    cf2 va[ITERS][2], vb[ITERS][2], w[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * NT;
        if (idx < TOT) {
            const int x2 = idx % LX2, k = idx / LX2, kb = k ? HH - k : 0;
            const cf2* L0 = nat + (2 * x2) * NPITCH; const cf2* L1 = L0 + NPITCH;
            va[it][0] = L0[k]; va[it][1] = L1[k]; vb[it][0] = L0[kb]; vb[it][1] = L1[kb]; w[it] = tw_full[k];
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * NT;
        if (idx < TOT) {
            const int x2 = idx % LX2, k = idx / LX2;
            float4* g = reinterpret_cast<float4*>(spec + x0 + 2 * x2);
            const size_t rk = (size_t)k * cols / 2, rh = (size_t)(HH - k) * cols / 2;      // rows in float4 units
            cf2 xk[2], xh[2];
            if (k == 0) {
#pragma unroll
                for (int l = 0; l < 2; ++l) { const cf2 z = va[it][l]; xk[l] = mk2(z.x + z.y, 0.f); xh[l] = mk2(z.x - z.y, 0.f); }
            } else {
#pragma unroll
                for (int l = 0; l < 2; ++l) r2c_pair(va[it][l], vb[it][l], w[it], xk[l], xh[l]);
            }
            g[rk] = make_float4(xk[0].x, xk[0].y, xk[1].x, xk[1].y);
            if (2 * k != HH) g[rh] = make_float4(xh[0].x, xh[0].y, xh[1].x, xh[1].y);
        }
    }
}
// transposed global load (k-major spectrum, 16 bytes per lane) -> c2r merge -> natural-order input of the packed
// inverse FFT in LDS.  All global loads of a thread are issued before the arithmetic.
template <class C>
__device__ __forceinline__ void a_load_pre(cf2* nat, const cf2* __restrict__ tw_full,
                                           const cf2* __restrict__ spec, int cols, int x0, int tid) {
    constexpr int HH = C::HALF, NPITCH = C::NPITCH, NT = C::NT, NP = HH / 2 + 1;
    constexpr int LX2 = C::LX / 2;
    constexpr int TOT = LX2 * NP, ITERS = (TOT + NT - 1) / NT;
    float4 va[ITERS], vb[ITERS]; cf2 w[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * NT;
        if (idx < TOT) {
            const int x2 = idx % LX2, k = idx / LX2;
            const float4* g = reinterpret_cast<const float4*>(spec + x0 + 2 * x2);
            va[it] = g[(size_t)k * cols / 2]; vb[it] = g[(size_t)(HH - k) * cols / 2]; w[it] = tw_full[k];
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int idx = tid + it * NT;
        if (idx < TOT) {
            const int x2 = idx % LX2, k = idx / LX2;
            cf2* L0 = nat + (2 * x2) * NPITCH; cf2* L1 = L0 + NPITCH;
            const cf2 xa[2] = { mk2(va[it].x, va[it].y), mk2(va[it].z, va[it].w) };
            const cf2 xb[2] = { mk2(vb[it].x, vb[it].y), mk2(vb[it].z, vb[it].w) };
            if (k == 0) {                                                     // imag of DC / Nyquist ignored (FFTW c2r)
                L0[0] = mk2(xa[0].x + xb[0].x, xa[0].x - xb[0].x);
                L1[0] = mk2(xa[1].x + xb[1].x, xa[1].x - xb[1].x);
            } else {
                cf2 zk[2], zh[2];
#pragma unroll
                for (int l = 0; l < 2; ++l) c2r_pair(xa[l], xb[l], w[it], zk[l], zh[l]);
                L0[k] = zk[0]; L1[k] = zk[1];
                if (2 * k != HH) { L0[HH - k] = zh[0]; L1[HH - k] = zh[1]; }
            }
        }
    }
}

#ifndef KCC_ROT_WPS
#define KCC_ROT_WPS 1
#endif
// 1: both horizontal taps of a sample row in one 2-byte LDS read (VERDICT r3 / r4 asked).  Measured in round 5: TWICE as slow
// (kA_fwd<240,rot8> 0.151 -> 0.301 ms, <360,rot8>/1280 0.300 -> 0.526): the 2-byte reads sit at arbitrary byte addresses and the
// LDS serves a misaligned ds_read_u16 far below the rate of two ds_read_u8.  Off.
#ifndef KCC_ROT8_U16
#define KCC_ROT8_U16 0
#endif
// SRC_ROT8 geometry: thread (line, j) of the first FFT pass owns dst rows 2(j + q MF), +1 for q < RF, i.e. q selects a band of
// BR = 2 MF dst rows.  The source of a band x 16-column block is a rotated rectangle; its bounding box (any angle) is at most
// BW x BH pixels (ceil(hypot(16, BR)) + alignment / tap margins; checked exhaustively over all 0.5-degree angles by
// tests/test_host_tables.py).  Every band is staged into its own fixed-size LDS box (row pitch PITCH bytes).
// middle segment size of the polar gather: the largest divisor of rf that is <= rf/4
__host__ __device__ constexpr int polar_qs_mid(int rf) { int d = rf / 4 > 0 ? rf / 4 : 1; while (rf % d) --d; return d; }
__host__ __device__ constexpr int isqrt_ceil(int v) { int r = 0; while (r * r < v) ++r; return r; }
template <int HH> struct Rot8Cfg {
    using D = Dir<typename FCfg<HH>::P, false>;
    static constexpr int BR = 2 * D::MF, NB = D::RF;
    static constexpr int DIAG = isqrt_ceil(FCfg<HH>::LX * FCfg<HH>::LX + BR * BR);
    static constexpr int BH = DIAG + 2;                       // rows of a box
    // 16-byte LDS-DMA pieces per box row (>= DIAG + 5 bytes: the box origin is aligned down to 4 pixels), forced ODD: a wave's
    // lanes read box rows two apart (dst rows 2j of consecutive j), i.e. 8*LPR dwords apart -- with an even LPR (64-byte rows)
    // that stride hits 1 or 2 of the 32 LDS banks (15x bank-conflict cycles at small angles, simulated), with an odd one 4
    static constexpr int LPR = ((DIAG + 5 + 15) / 16) | 1;
    static constexpr int PITCH = LPR * 16;                    // bytes per box row
    static constexpr int BOX = BH * PITCH;
    static constexpr int XY_OFF = ((NB * BOX + 15) / 16) * 16;        // [X0 2HH | Y0 2HH] ints
    static constexpr int INFO_OFF = XY_OFF + 16 * HH;                 // int2 (ox, oy) per band
    static constexpr int BYTES_ = INFO_OFF + 8 * NB;
    static constexpr size_t BYTES = BYTES_ > (int)FCfg<HH>::BYTES ? (size_t)BYTES_ : FCfg<HH>::BYTES;
};
template <int HH, int SRC> __host__ __device__ constexpr size_t fwd_lds_bytes() {
    return SRC == SRC_ROT8 ? Rot8Cfg<HH>::BYTES : FCfg<HH>::BYTES;
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int HH, int SRC>
__global__ __launch_bounds__(FCfg<HH>::NT, ((SRC == SRC_ROT || SRC == SRC_ROT8) ? KCC_ROT_WPS : FCfg<HH>::WPS)) void kA_fwd(AArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = FCfg<HH>; using P = typename C::P; using D = Dir<P, false>;
    constexpr bool WL = KCC_WAVE_LOCAL && (64 % C::T == 0);   // lines are wave-local: no workgroup barriers inside the chain
    constexpr bool POLAR = (SRC == SRC_POLAR_H || SRC == SRC_POLAR_Q || SRC == SRC_POLAR_T);
    cf2* lds = reinterpret_cast<cf2*>(smem);
    if (ABL(a, 8)) return;
    const int tid = threadIdx.x, line = tid / C::T, j = tid - line * C::T;
    int bx, item;
    constexpr int A_LX = C::LX;
    if (POLAR) {
        // tile-major order, outermost (most expensive) ring first: the workgroups running at any time share one tile's
        // gather tables (L2-resident), and every item's annulus is read exactly once
        // ... in groups of a.polar_group neighbouring tiles: a tile's ring is only ~8 source pixels thick, so a 128-byte line of
        // the source plane (32 pixels of a column) serves three or four neighbouring tiles -- run them back to back on the XCD
        // that holds the line (the group's tables, polar_group x 92 KB, still fit its L2)
        const int nbx = a.cols / A_LX, G = a.polar_group, per = G * a.n_items, per8 = (per + 7) / 8 * 8;
        const int g = (int)blockIdx.x / per8, loc = (int)blockIdx.x - g * per8;
        if (loc >= per) return;                              // padding of the group to a multiple of 8 blocks
        int tg;
        xcd_coords_of(loc, G, a.n_items, tg, item, a.rev);
        bx = nbx - 1 - (g * G + tg);
    } else {
        xcd_coords(a.cols / A_LX, a.n_items, bx, item, a.rev);
    }
    const int x0 = bx * A_LX;

    cf2 vin[1][D::RF], vout[1][D::RL];
#ifdef KCC_ABLATE
    zero_fill(vin[0]);
#endif
    if (SRC == SRC_PLANE) {
        if (j < D::MF) {
            const int pl = a.src_idx ? a.src_idx[item] : item;
            const cf2* src = reinterpret_cast<const cf2*>(a.src + (size_t)pl * a.src_stride + (size_t)(x0 + line) * a.src_pitch);
#pragma unroll
            for (int q = 0; q < D::RF; ++q) vin[0][q] = src[j + q * D::MF];
        }
    }
    if (POLAR && !ABL(a, 1)) {
        // polar(fftshift(RemoveZeroComponent(p)))  (correlation_flow.cc:228-236).  The source pixels of the tile's
        // samples (A_LX radii x all angles = an annulus) are staged in LDS one angular segment at a time: the host lists,
        // per tile and segment, the runs of source pixels as chunks of 16 consecutive floats (column-major S, so a
        // chunk is a piece of one source column); chunk c lands in LDS floats [16c, 16c+16) via LDS-DMA (four lanes x
        // 16 bytes).  Every thread then bilinearly samples its own first-pass
        // points straight into registers from entries that hold the LDS positions of the two tap columns.
        constexpr int QS = SRC == SRC_POLAR_H ? D::RF / 2 : (SRC == SRC_POLAR_T ? polar_qs_mid(D::RF) : 1), NSEG = D::RF / QS;
        static_assert(NSEG * QS == D::RF, "segments must tile the first-pass points");
        const float* S = a.src + (size_t)item * a.src_stride;
        float* ldsf = reinterpret_cast<float*>(smem);
        const uint4* pts = a.polar_pts + (size_t)bx * D::RF * C::NT + tid;
        const int l4 = tid & 3;
#pragma unroll
        for (int seg = 0; seg < NSEG; ++seg) {
            if (seg) __syncthreads();                        // previous segment consumed
            const int c0 = a.polar_seg_first[bx * NSEG + seg], nch = a.polar_seg_first[bx * NSEG + seg + 1] - c0;
            const uint32_t* ch = a.polar_chunks + c0;
            // (chunk offsets first, LDS-DMA afterwards: a wait for an offset would also wait for every DMA issued before it)
            constexpr int G = 8, PER = C::NT / 4;            // chunks per pass of the workgroup; G passes per group
            if (!ABL(a, 16))
            for (int cb = 0; cb < nch; cb += G * PER) {
                unsigned off[G];
#pragma unroll
                for (int g = 0; g < G; ++g) { const int c = cb + g * PER + (tid >> 2); off[g] = (c < nch ? ch[c] : 0u) + 4 * l4; }
#pragma unroll
                for (int g = 0; g < G; ++g) asm volatile("" : "+v"(off[g]));     // the offsets are consumed (waited for) HERE
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int c = cb + g * PER + (tid >> 2);
                    if (c < nch)
                        __builtin_amdgcn_global_load_lds((glb_void*)(S + off[g]), (lds_void*)(ldsf + (cb + g * PER) * 16 + (tid & ~63) * 4), 16, 0, 0);
                }
            }
            uint4 e[QS];
            if (j < D::MF) {
#ifdef KCC_UB_POLAR8
                const uint2* pts2 = reinterpret_cast<const uint2*>(a.polar_pts) + (size_t)bx * D::RF * C::NT + tid;
#pragma unroll
                for (int qq = 0; qq < QS; ++qq) { const uint2 h2 = pts2[(size_t)(seg * QS + qq) * C::NT]; e[qq] = make_uint4(h2.x, h2.y & 0x7FFFu, h2.x + 2u, (h2.y & 0x7FFFu) + 2u); }
#else
#pragma unroll
                for (int qq = 0; qq < QS; ++qq) e[qq] = pts[(size_t)(seg * QS + qq) * C::NT];
#endif
            }
            __syncthreads();
            if (j < D::MF && !ABL(a, 32)) {
#pragma unroll
                for (int qq = 0; qq < QS; ++qq) {
                    float r[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const unsigned lo = h ? e[qq].z : e[qq].x, hi = h ? e[qq].w : e[qq].y;
                        const cf2 va = load2(ldsf + (lo & 0xFFFFu)), vb = load2(ldsf + hi);   // (sx; sy, sy+1), (sx+1; sy, sy+1)
                        r[h] = bilerp(va.x, vb.x, va.y, vb.y, (lo >> 16) & 31, (lo >> 21) & 31);
                    }
                    vin[0][seg * QS + qq] = mk2(r[0], r[1]);
                }
            }
        }
        __syncthreads();                                     // staged pixels consumed before the exchange overwrites them
    }
    if (SRC == SRC_ROT && !ABL(a, 1)) {
        // stage this angle's row terms X0[r], Y0[r] (2H ints) in LDS: the gather then has ONE dependent global
        // stage (the taps) instead of two (table, then taps)
        const int* tab = a.rot_tab + (size_t)a.rot_index[item] * (2 * a.cols + 2 * a.rows);
        int* xy = reinterpret_cast<int*>(lds);               // [X0 H | Y0 H], overwritten by the exchange afterwards
        constexpr int NXY = 4 * HH;                          // 2 * rows; compile-time trip count: all loads issued together
#pragma unroll
        for (int it = 0; it < (NXY + C::NT - 1) / C::NT; ++it) {
            const int i = tid + it * C::NT;
            if (i < NXY) xy[i] = tab[2 * a.cols + i];
        }
        const int c = x0 + line;
        const int ad = tab[c], bd = tab[a.cols + c];
        __syncthreads();
        if (j < D::MF) {
            // (the slot index is the same for the whole workgroup: keep the image base in SGPRs)
            const float* img = a.src + (size_t)__builtin_amdgcn_readfirstlane(a.src_idx[item]) * a.src_stride;
            const int2* X0 = reinterpret_cast<const int2*>(xy);
            const int2* Y0 = reinterpret_cast<const int2*>(xy + a.rows);
#pragma unroll
            for (int q = 0; q < D::RF; ++q) {
                const int m = j + q * D::MF;
                const int2 xr = X0[m], yr = Y0[m];
                vin[0][q] = mk2(rot_sample(img, a.rows, a.src_pitch, a.cols, ad, bd, xr.x, yr.x),
                                        rot_sample(img, a.rows, a.src_pitch, a.cols, ad, bd, xr.y, yr.y));
                // cap the number of gathers in flight (register pressure -> occupancy): no hoisting across groups
                if ((q % KCC_GATHER_GROUP) == KCC_GATHER_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                     // table consumed before the exchange buffer is written
    }
    if (SRC == SRC_ROT8 && !ABL(a, 1)) {
        // RotateArray (utils.cc:154-161) = cv::warpAffine(INTER_LINEAR, BORDER_WRAP) from the u8 frame store.  Band q of
        // the tile (dst rows [q BR, (q+1) BR) x 16 dst columns) reads a rotated rectangle of the source; its bounding box
        // is staged in LDS (row-major bytes, BORDER_WRAP applied while staging: rows modulo H, columns through the 16
        // repeated columns behind every stored row), so the taps are plain LDS byte reads at un-wrapped coordinates.
        using R = Rot8Cfg<HH>;
        const int rows = 2 * HH, W = a.cols;
        const int* tab = a.rot_tab + (size_t)a.rot_index[item] * (2 * W + 2 * rows);
        int* xy = reinterpret_cast<int*>(smem + R::XY_OFF);
        int2* info = reinterpret_cast<int2*>(smem + R::INFO_OFF);
        constexpr int NXY = 4 * HH;
#pragma unroll
        for (int it = 0; it < (NXY + C::NT - 1) / C::NT; ++it) {
            const int i = tid + it * C::NT;
            if (i < NXY) xy[i] = tab[2 * W + i];
        }
        const int c = x0 + line;
        const int ad = tab[c], bd = tab[W + c];
        if (tid < R::NB) {
            // box origin of band `tid`: the fixed-point coordinate terms are monotone in r and in c, so the extremes sit
            // at the corners of the block
            const int r0 = tid * R::BR, r1 = r0 + R::BR - 1;
            const int ax0 = tab[x0], ax1 = tab[x0 + A_LX - 1], ay0 = tab[W + x0], ay1 = tab[W + x0 + A_LX - 1];
            const int X0a = tab[2 * W + r0], X0b = tab[2 * W + r1], Y0a = tab[2 * W + rows + r0], Y0b = tab[2 * W + rows + r1];
            const int xmin = (min(X0a, X0b) + min(ax0, ax1)) >> 10, ymin = (min(Y0a, Y0b) + min(ay0, ay1)) >> 10;
            info[tid] = make_int2(xmin & ~3, ymin);
        }
        __syncthreads();
        const uint8_t* img = a.src8 + (size_t)__builtin_amdgcn_readfirstlane(a.src_idx[item]) * a.src8_stride;
        constexpr int SLOTS = R::NB * R::BH * R::LPR;        // 16-byte pieces of all boxes; piece i lands at LDS byte 16 i
        constexpr int ITERS = (SLOTS + C::NT - 1) / C::NT;
        // (all source addresses first -- they need the box origins from LDS -- then the LDS-DMAs back to back: an LDS read
        // after a DMA makes the compiler wait for that DMA)
        const uint8_t* from[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int i = min(tid + it * C::NT, SLOTS - 1);
            const int b = i / (R::BH * R::LPR), rem = i - b * (R::BH * R::LPR), k = rem / R::LPR, p = rem - k * R::LPR;
            const int2 o = info[b];
            int y = o.y + k; y += (y < 0) ? rows : 0; y -= (y >= rows) ? rows : 0;
            int x = o.x + 16 * p; x += (x < 0) ? W : 0; x -= (x >= W) ? W : 0;
            from[it] = img + (unsigned)(y * a.src8_pitch + x);
        }
        __builtin_amdgcn_sched_barrier(0);                   // keep every LDS read above the first DMA
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (tid + it * C::NT < SLOTS && !ABL(a, 16))
                __builtin_amdgcn_global_load_lds((glb_void*)from[it], (lds_void*)(smem + (size_t)(it * C::NT + (tid & ~63)) * 16), 16, 0, 0);
        }
        __syncthreads();
        if (j < D::MF && !ABL(a, 32)) {
            const int2* X0 = reinterpret_cast<const int2*>(xy);
            const int2* Y0 = reinterpret_cast<const int2*>(xy + rows);
            const uint8_t* sb = reinterpret_cast<const uint8_t*>(smem);
#pragma unroll
            for (int q = 0; q < D::RF; ++q) {
                const int m = j + q * D::MF;
                const int2 xr = X0[m], yr = Y0[m];
                const int2 o = info[q];
                const uint8_t* box = sb + q * R::BOX - o.y * R::PITCH - o.x;
                float r[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int X = ((h ? xr.y : xr.x) + ad) >> 5, Y = ((h ? yr.y : yr.x) + bd) >> 5;
                    const uint8_t* t = box + (Y >> 5) * R::PITCH + (X >> 5);
#if KCC_ROT8_U16
                    // both horizontal taps of a row in ONE LDS read (a 2-byte read at any byte address); the bytes go straight into
                    // v_cvt_f32_ubyte0 / ubyte1 -- half the LDS instructions of the sampling, same values, twice the time (above)
                    typedef unsigned short u16u __attribute__((aligned(1)));
                    const unsigned ta = *reinterpret_cast<const u16u*>(t), tb = *reinterpret_cast<const u16u*>(t + R::PITCH);
                    r[h] = bilerp(unit_u8(ta & 255u), unit_u8(ta >> 8), unit_u8(tb & 255u), unit_u8(tb >> 8), X & 31, Y & 31);
#else
                    r[h] = bilerp(unit_u8(t[0]), unit_u8(t[1]), unit_u8(t[R::PITCH]), unit_u8(t[R::PITCH + 1]), X & 31, Y & 31);
#endif
                }
                vin[0][q] = mk2(r[0], r[1]);
            }
        }
        __syncthreads();                                     // boxes consumed before the exchange buffer is written
    }
    if (a.dbg_plane && j < D::MF) {                          // debug tap: what the gather produced (tests: bit-exact vs the oracle)
        cf2* o = reinterpret_cast<cf2*>(a.dbg_plane + ((size_t)item * a.cols + x0 + line) * (size_t)(2 * HH));
#pragma unroll
        for (int q = 0; q < D::RF; ++q) o[j + q * D::MF] = vin[0][q];
    }
    cf2* const ex[1] = { C::ex_of(lds, line) };
    if (!ABL(a, 4)) fft_chain<P, false, 1, WL>(vin, vout, j, ex, a.tw_f);
    region_sync<C::WREG>();                                  // exchange buffer fully consumed
    if (j < D::ML) {
#pragma unroll
        for (int q = 0; q < D::RL; ++q) lds[line * C::NPITCH + j + q * D::ML] = vout[0][q];
    }
    __syncthreads();
    if (!ABL(a, 2)) a_post_store<C>(lds, a.tw_full, a.spec + (size_t)item * a.spec_stride, a.cols, x0, tid);
}

// ConvertMatToNormalizedArray (utils.cc:110-118) fused into the first FFT pass, persistent form: a workgroup transforms
// `tpw` consecutive 16-column tiles of ONE image and fetches the rows of the next tile (two 16-byte loads per thread, held in
// registers) while it transforms the current one, so the load latency -- which dominated the one-tile-per-workgroup form --
// hides behind the FFT (measured: 0.137 -> 0.119 ms at two tiles per workgroup; more tiles per workgroup leave too few
// workgroups).  The tile's 16 image columns are 16 bytes of every image row: rows are staged in LDS ([row][16 bytes])
// and copied into the u8 frame store (the de-rotation of ComputePose reads the image from there); a thread then picks up
// its first-pass points as bytes.
#ifndef KCC_U8_TPW
#define KCC_U8_TPW 2
#endif
// 1: file the image in the u8 frame store by whole rows (full-line writes) instead of tile by tile.  Measured (round 4): the
// kernel's write traffic falls to its nominal bytes, its TIME rises -- 0.119 -> 0.127 ms at 640x480, 0.316 -> 0.327 ms at
// 1280x720 (the extra row loads and stores sit in front of the first tile's transform) -- so the tile-wise copy stays.
#ifndef KCC_U8_COPY_ROWS
#define KCC_U8_COPY_ROWS 0
#endif
#if !defined(KCC_ABLATE) && (KCC_ROT8_U16 || KCC_U8_COPY_ROWS)
#error "KCC_ROT8_U16 / KCC_U8_COPY_ROWS are measured no-go forms: tuning library (-DKCC_ABLATE) only"
#endif
template <int HH>
// (the register prefetch of the next tile needs ~123 VGPRs at 240 points and ~150 at 360: never ask for more waves per
// SIMD than that leaves room for -- a 128-register cap made the 360-point kernel spill 33 dwords: 0.396 -> 0.331 ms at HD)
#ifndef KCC_U8_WPS
#define KCC_U8_WPS(hh) ((hh) > 240 ? 3 : 4)
#endif
__global__ __launch_bounds__(FCfg<HH>::NT, (FCfg<HH>::WPS > KCC_U8_WPS(HH) ? KCC_U8_WPS(HH) : FCfg<HH>::WPS)) void kA_fwd_u8(AArgs a, int tpw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = FCfg<HH>; using P = typename C::P; using D = Dir<P, false>;
    constexpr bool WL = KCC_WAVE_LOCAL && (64 % C::T == 0);
    constexpr int A_LX = C::LX, ROWS = 2 * HH, NR = (ROWS + C::NT - 1) / C::NT;
    cf2* lds = reinterpret_cast<cf2*>(smem);
    if (ABL(a, 8)) return;
    int g, item;
    xcd_coords((a.cols / A_LX) / tpw, a.n_items, g, item, a.rev);           // groups of tpw tiles; an image's groups share an XCD
    const uint8_t* in = a.src8 + (size_t)item * a.src8_stride;
    uint8_t* keep = a.dst8 ? a.dst8 + (size_t)a.dst8_slot[item] * a.dst8_stride : nullptr;
    uint4 cur[NR], nxt[NR];
    auto fetch = [&](uint4 (&v)[NR], int x0, int tid) {
#pragma unroll
        for (int it = 0; it < NR; ++it) {
            const int r = tid + it * C::NT;
            if (r < ROWS && !ABL(a, 1)) v[it] = *reinterpret_cast<const uint4*>(in + (size_t)r * a.src8_pitch + x0);
            else v[it] = make_uint4(0, 0, 0, 0);
        }
    };
    fetch(cur, g * tpw * A_LX, threadIdx.x);
#if KCC_U8_COPY_ROWS
    // The image's copy in the u8 frame store (the de-rotation of ComputePose reads it from there) is written by ROWS, not by
    // tiles: the workgroups of an image share its rows out, each copies whole rows in 16-byte pieces that are contiguous
    // across the lanes -- full-line writes.  (Filed tile by tile -- 16 bytes of every row per tile -- the stores were partial
    // lines: the u8 kernel wrote 1.45x its nominal bytes)  The source rows are the ones the image's tiles are
    // reading anyway on the same XCD: the second read is an L2 hit.
    if (keep && !ABL(a, 64)) {
        const int ng = (a.cols / A_LX) / tpw, per = (ROWS + ng - 1) / ng, r0 = g * per, r1 = min(ROWS, r0 + per);
        const int cpr = a.cols / 16 + 1;                      // 16-byte pieces per row: W / 16 of the image + the wrap columns
        const int total = (r1 - r0) * cpr;
        for (int i = (int)threadIdx.x; i < total; i += C::NT) {
            const int r = r0 + i / cpr, ch = i % cpr;
            const uint4 v = *reinterpret_cast<const uint4*>(in + (size_t)r * a.src8_pitch + (ch == cpr - 1 ? 0 : 16 * ch));
            *reinterpret_cast<uint4*>(keep + (size_t)r * a.dst8_pitch + 16 * ch) = v;
        }
    }
#endif
    for (int t = 0; t < tpw; ++t) {
        const int bx = g * tpw + t, x0 = bx * A_LX;
        // Everything per-thread is re-derived per tile from an opaque copy of the thread index, and the twiddle tables are
        // re-read: hoisted out of the loop (which the compiler does eagerly) those values sit in ~60 VGPRs for the loop's
        // whole length and push the kernel into scratch spills.
        int tid = threadIdx.x;
        const cf2* tw_f = a.tw_f; const cf2* tw_full = a.tw_full;
        asm volatile("" : "+v"(tid), "+s"(tw_f), "+s"(tw_full));
        const int line = tid / C::T, j = tid - line * C::T;
        if (t + 1 < tpw) fetch(nxt, x0 + A_LX, tid);
        uint4* st = reinterpret_cast<uint4*>(smem);
#pragma unroll
        for (int it = 0; it < NR; ++it) {
            const int r = tid + it * C::NT;
            if (r < ROWS) {
                st[r] = cur[it];
                if (!KCC_U8_COPY_ROWS && keep && !ABL(a, 64)) {
                    *reinterpret_cast<uint4*>(keep + (size_t)r * a.dst8_pitch + x0) = cur[it];
                    if (bx == 0) *reinterpret_cast<uint4*>(keep + (size_t)r * a.dst8_pitch + a.cols) = cur[it];   // wrap columns
                }
            }
        }
        __syncthreads();
        cf2 vin[1][D::RF], vout[1][D::RL];
        if (j < D::MF) {
            const uint8_t* sb = reinterpret_cast<const uint8_t*>(smem) + line;
#pragma unroll
            for (int q = 0; q < D::RF; ++q) {
                const int m = j + q * D::MF;
                vin[0][q] = mk2(unit_u8(sb[32 * m]), unit_u8(sb[32 * m + 16]));
            }
        }
        __syncthreads();                                     // staged rows consumed before the exchange overwrites them
        cf2* const ex[1] = { C::ex_of(lds, line) };
        if (!ABL(a, 4)) fft_chain<P, false, 1, WL>(vin, vout, j, ex, tw_f);
        region_sync<C::WREG>();                              // exchange buffer fully consumed
        if (j < D::ML) {
#pragma unroll
            for (int q = 0; q < D::RL; ++q) lds[line * C::NPITCH + j + q * D::ML] = vout[0][q];
        }
        __syncthreads();
        if (!ABL(a, 2)) a_post_store<C>(lds, tw_full, a.spec + (size_t)item * a.spec_stride, a.cols, x0, tid);
        __syncthreads();                                     // natural buffer consumed before the next tile is staged
#pragma unroll
        for (int it = 0; it < NR; ++it) cur[it] = nxt[it];
    }
}

// row r within `radius` (cyclically) of the window centre, or -- rotation surfaces, whose source is point-symmetric --
// of the centre's 180-degree mirror row
__device__ __forceinline__ bool win_hit(int r, int centre, int rows, int radius, int mirror) {
    int d = abs(r - centre); d = min(d, rows - d);
    if (mirror) { int m = abs(d - rows / 2); d = min(d, m); }
    return d <= radius;
}

template <int HH, int EPI, int LXO = 0>
__global__ __launch_bounds__((ICfg<HH, EPI, LXO>::NT), (ICfg<HH, EPI, LXO>::WPS)) void kA_inv(AArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = ICfg<HH, EPI, LXO>; using P = typename C::P; using DI = Dir<P, true>; using DF = Dir<P, false>;
    constexpr bool WL = KCC_WAVE_LOCAL && (64 % C::T == 0);
    constexpr int NW = (C::NT + 63) / 64;
    cf2* lds = reinterpret_cast<cf2*>(smem);
    if (ABL(a, 8)) return;
    __shared__ float red_f[NW];
    __shared__ int red_i[NW];
    __shared__ double red_d[2][NW];
    const int tid = threadIdx.x, line = tid / C::T, j = tid - line * C::T;
    constexpr int A_LX = C::LX;
    const int nbx = a.cols / A_LX;
    int bx, item2;
    constexpr bool KFWD = epi_is_kfwd(EPI);
    constexpr int KT = EPI == EPI_KFWD_POLY3 ? KT_POLY3 : (EPI == EPI_KFWD_POLYN ? KT_POLYN : KT_GAUSS);
    int item, plane;
    if (KFWD && a.zz_tiles > 0) {
        // Kzz's kernel plane is real and even (an autocorrelation), so its column W-c is the y-reversed column c and
        // transforms to the conjugate: only the columns <= W/2 of plane 0 are processed (solve_inv mirrors the rest)
        int t;
        xcd_coords(a.zz_tiles + nbx, a.n_items, t, item, a.rev);
        plane = t < a.zz_tiles ? 0 : 1;
        bx = plane ? t - a.zz_tiles : t;
    } else if (EPI == EPI_SHIFTED && a.zz_tiles > 0) {
        // IFFT(|F|) is real and even as well: only the column tiles covering [0, W/2] are transformed, each column is
        // also written to its mirror position (see the epilogue)
        xcd_coords(a.zz_tiles, a.n_items, bx, item, a.rev);
        plane = 0;
    } else {
        xcd_coords(nbx, a.n_items * (KFWD ? a.n_planes : 1), bx, item2, a.rev);
        item = KFWD ? item2 / a.n_planes : item2;
        plane = KFWD ? a.plane_first + item2 % a.n_planes : 0;
    }
    const int x0 = bx * A_LX;
    cf2* spec = a.spec + (size_t)item * a.spec_stride + (size_t)plane * a.plane_stride;

    if (!ABL(a, 1)) a_load_pre<C>(lds, a.tw_full, spec, a.cols, x0, tid);
    __syncthreads();
    cf2 vin[1][DI::RF], vout[1][DI::RL];
    if (j < DI::MF) {
#pragma unroll
        for (int q = 0; q < DI::RF; ++q) vin[0][q] = lds[line * C::NPITCH + j + q * DI::MF];
    }
    region_sync<C::WREG>();                                  // natural buffer consumed before the exchange overwrites it
    cf2* const ex[1] = { C::ex_of(lds, line) };
    if (!ABL(a, 4)) fft_chain<P, true, 1, WL>(vin, vout, j, ex, a.tw_i);
    const float size = (float)((long)a.rows * a.cols);       // IFFT: x / x.size()  (correlation_flow.cc:76)
    const float rsize = 1.0f / size;                          // (applied as a multiplication: 1 ulp, far below FFT rounding)

    if (EPI == EPI_SHIFTED) {
        // write fftshift(p) into the zero-bordered plane S (column pitch rows+2): out(r,c) at ((r+H/2)%H, (c+W/2)%W)
        // (circ_shift.h:238-244).  RemoveZeroComponent (correlation_flow.cc:79-87) is applied on the way when the launcher asks
        // for it (fix_zero; needs the mirrored half-plane form): column 0 becomes (p(r,1) + p(r,W-1))/2 from the ORIGINAL
        // columns -- p(r,W-1) is p(-r,1), and column 1 sits in the same tile as column 0 --, then row 0 of every other column
        // becomes (p(1,c) + p(H-1,c))/2: the values k_fix_zero computes from the stored plane, without its launch.
        const int xcol = x0 + line;
        const bool half = a.zz_tiles > 0;                    // columns > W/2 come from their mirrors
        constexpr int H = 2 * HH, SP = H + 2, HQ = H / 2;
        float v0[DI::RL], v1[DI::RL];
#pragma unroll
        for (int q = 0; q < DI::RL; ++q) { v0[q] = vout[0][q].x * rsize; v1[q] = vout[0][q].y * rsize; }
        if (a.fix_zero && half) {                            // (uniform)
            float* sf = reinterpret_cast<float*>(lds);       // [0, H): column 1;  [H, H + LX): row H-1 of every column of the tile
            __syncthreads();                                 // exchange buffers consumed
            if (j < DI::ML) {
                if (xcol == 1) {
#pragma unroll
                    for (int q = 0; q < DI::RL; ++q) { const int r = 2 * j + 2 * q * DI::ML; sf[r] = v0[q]; sf[r + 1] = v1[q]; }
                }
                if (j == DI::ML - 1) sf[H + line] = v1[DI::RL - 1];
            }
            __syncthreads();
            if (j < DI::ML) {
                if (xcol == 0) {
#pragma unroll
                    for (int q = 0; q < DI::RL; ++q) {
                        const int r = 2 * j + 2 * q * DI::ML;
                        v0[q] = (sf[r] + sf[r == 0 ? 0 : H - r]) * 0.5f;
                        v1[q] = (sf[r + 1] + sf[H - r - 1]) * 0.5f;
                    }
                } else if (j == 0) {
                    v0[0] = (v1[0] + sf[H + line]) * 0.5f;
                }
            }
        }
        if (j < DI::ML && !(half && xcol > a.cols / 2)) {
            // H is the template's: with 0 <= j < ML known, the cyclic wraps below fold to constants for all but the one or
            // two q whose rows straddle H/2 (this epilogue was 60 % integer work with a run-time H)
            __builtin_assume(j >= 0); __builtin_assume(j < DI::ML);
            const int W = a.cols;
            int xs = xcol + W / 2; if (xs >= W) xs -= W;
            float* col = a.real_out + (size_t)item * a.real_stride + (size_t)xs * SP;
            // p(r, c) = p(-r, -c): the same values are column W - c read backwards
            const bool mirror = half && xcol != 0 && 2 * xcol != W;
            int xm = W - xcol + W / 2; if (xm >= W) xm -= W;
            float* colm = a.real_out + (size_t)item * a.real_stride + (size_t)xm * SP;
            typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
            for (int q = 0; q < DI::RL; ++q) {
                const int r = 2 * j + 2 * q * DI::ML;
                const int ys = r >= HQ ? r - HQ : r + HQ;
                if (HQ & 1) { col[ys] = v0[q]; col[ys + 1 >= H ? ys + 1 - H : ys + 1] = v1[q]; }
                else *reinterpret_cast<cf2*>(col + ys) = mk2(v0[q], v1[q]);
                if (mirror) {
                    // rows -r and -(r+1) sit at y0 = (HQ - r) mod H and y0 - 1: neighbours in memory except across the
                    // cyclic seam (r == HQ), so one 8-byte store
                    const int y0 = r > HQ ? 3 * HQ - r : HQ - r;
                    if (r == HQ) { colm[0] = v0[q]; colm[H - 1] = v1[q]; }
                    else { f2u pr; pr.x = v1[q]; pr.y = v0[q]; *reinterpret_cast<f2u*>(colm + y0 - 1) = pr; }
                }
            }
        }
    } else if (EPI == EPI_REAL) {
        if (j < DI::ML) {
            cf2* dst = reinterpret_cast<cf2*>(a.real_out + (size_t)item * a.real_stride + (size_t)(x0 + line) * a.rows);
#pragma unroll
            for (int q = 0; q < DI::RL; ++q) dst[j + q * DI::ML] = mk2(vout[0][q].x * rsize, vout[0][q].y * rsize);
        }
    } else if (KFWD) {
        static_assert(DI::RL == DF::RF && DI::ML == DF::MF, "inverse output layout must equal forward input layout");
        float gbias = 0.f, gscale = 0.f;
        if (KT == KT_GAUSS) {
            const float xx = a.energy[2 * item + 0] / size, zz = a.energy[2 * item + 1] / size;
            gbias = (plane == 0) ? (zz + zz) : (xx + zz);
            gscale = (-1.f / (a.fn.sigma * a.fn.sigma)) / size;
        }
        float mx = 0.f;
        if (j < DI::ML) {
#pragma unroll
            for (int q = 0; q < DI::RL; ++q) {
                const float k0 = kernel_value<KT>(a.fn, vout[0][q].x * rsize, gbias, gscale);
                const float k1 = kernel_value<KT>(a.fn, vout[0][q].y * rsize, gbias, gscale);
                mx = fmaxf(mx, fmaxf(fabsf(k0), fabsf(k1)));
                vout[0][q] = mk2(k0, k1);
            }
        }
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        // max|k| of the plane (k /= max, correlation_flow.cc:214): every wave files the max of its own lines as one part --
        // a plain store, no atomic and no workgroup rendezvous; the ridge solve folds the parts (parts_max)
        if ((tid & 63) == 0) a.maxbuf[(size_t)(2 * item + plane) * KCC_MAXPARTS + bx * NW + (tid >> 6)] = __float_as_uint(mx);
        region_sync<C::WREG>();                              // exchange buffer consumed before the forward chain
        cf2 fout[1][DF::RL];
        if (!ABL(a, 4)) fft_chain<P, false, 1, WL>(vout, fout, j, ex, a.tw_f);
        region_sync<C::WREG>();
        if (j < DF::ML) {
#pragma unroll
            for (int q = 0; q < DF::RL; ++q) lds[line * C::NPITCH + j + q * DF::ML] = fout[0][q];
        }
        __syncthreads();
        if (!ABL(a, 2)) a_post_store<C>(lds, a.tw_full, spec, a.cols, x0, tid);
    } else {
        // arg-max (column-major first strict max, Eigen maxCoeff visitor) + moments for GetInfo
        float best = -INFINITY; int bidx = 0x7FFFFFFF;
        float s1 = 0.f, s2 = 0.f;
        int wr = 0; bool in_col = true;
        if (EPI == EPI_ARGMAX_WIN) {
            wr = a.win_row[item];
            int dc = abs(x0 + line - a.win_col[item]); dc = min(dc, a.cols - dc);
            in_col = dc <= a.win_radius;
        }
        if (j < DI::ML) {
            const int base = (x0 + line) * a.rows;
#pragma unroll
            for (int q = 0; q < DI::RL; ++q) {               // increasing q -> increasing linear index
                const float g0 = vout[0][q].x * rsize, g1 = vout[0][q].y * rsize;
                const int li = base + 2 * (j + q * DI::ML);
                if (EPI == EPI_ARGMAX_WIN) {
                    // candidates only inside the (2R+1)^2 cyclic window (the moments still cover the whole surface)
                    const int r0 = 2 * (j + q * DI::ML);
                    if (in_col && win_hit(r0, wr, a.rows, a.win_radius, a.win_mirror) && g0 > best) { best = g0; bidx = li; }
                    if (in_col && win_hit(r0 + 1, wr, a.rows, a.win_radius, a.win_mirror) && g1 > best) { best = g1; bidx = li + 1; }
                } else {
                    if (g0 > best) { best = g0; bidx = li; }
                    if (g1 > best) { best = g1; bidx = li + 1; }
                }
#ifndef KCC_UB_NOMOMENTS
                s1 += g0 + g1; s2 += g0 * g0 + g1 * g1;
#endif
            }
        }
        double d1 = (double)s1, d2 = (double)s2;
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best, off); const int oi = __shfl_xor(bidx, off);
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
            d1 += __shfl_xor(d1, off); d2 += __shfl_xor(d2, off);
        }
        if ((tid & 63) == 0) { red_f[tid >> 6] = best; red_i[tid >> 6] = bidx; red_d[0][tid >> 6] = d1; red_d[1][tid >> 6] = d2; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NW; ++w) {
                if (red_f[w] > best || (red_f[w] == best && red_i[w] < bidx)) { best = red_f[w]; bidx = red_i[w]; }
                d1 += red_d[0][w]; d2 += red_d[1][w];
            }
            Partial p; p.sum = d1; p.sumsq = d2; p.peak = best; p.idx = bidx;
            a.partials[(size_t)item * a.partial_stride + bx] = p;
        }
    }
}

// geometry of the forward A kernels for half length hh: lines per tile, threads per line, first-pass radix and stride
// (thread (line, j), j < mf, owns points j + q*mf, q < rf), LDS bytes of the FFT buffers (the polar staging must fit
// under max(this, its own size)); the host builds the polar gather tables from it
FwdGeom fwd_geom(int hh) {
    FwdGeom f{};
#define X(n) if (hh == n) { using C = FCfg<n>; using D = Dir<C::P, false>; f.lines = C::LX; f.threads = C::T; f.rf = D::RF; f.mf = D::MF; f.lds_bytes = C::BYTES; \
                      f.qs_opts[0] = D::RF % 2 == 0 ? D::RF / 2 : 1; f.qs_opts[1] = polar_qs_mid(D::RF); f.qs_opts[2] = 1; }
    KCC_HALF_LIST(X)
#undef X
    return f;
}
// SRC_ROT8 box geometry (tests check the bounds exhaustively on the host)
Rot8Geom rot8_geom(int hh) {
    Rot8Geom r{};
#define X(n) if (hh == n) { using R = Rot8Cfg<n>; r.band_rows = R::BR, r.bands = R::NB; r.box_rows = R::BH; r.pitch = R::PITCH; r.lds_bytes = (int)R::BYTES; }
    KCC_HALF_LIST(X)
#undef X
    return r;
}
// columns [0, W/2] rounded up to whole kernel_fwd tiles: what the Hermitian-half Kzz transform reads of the zz plane
int zz_half_columns(PlaneGeom g) { const int lx = inv_lx_for(g.rows / 2, g.cols, EPI_KFWD_POLY3); return std::min(g.cols, ((g.cols / 2) / lx + 1) * lx); }
// columns [0, min(W/2, need)] rounded up to whole tiles: what the even-half inverse row pass of the zero-phase image reads
// when only its columns |c| <= need are consumed (the polar gather never leaves the inscribed circle)
int shifted_columns(PlaneGeom g, int need) { const int lx = inv_lx_for(g.rows / 2, g.cols, EPI_SHIFTED); return std::min(g.cols, (std::min(g.cols / 2, need) / lx + 1) * lx); }
int argmax_blocks(PlaneGeom g) { return g.cols / inv_lx_for(g.rows / 2, g.cols, EPI_ARGMAX); }
// running-max parts kernel_fwd files per kernel plane: one per wave and column tile (half: the Hermitian-half zz plane)
int kfwd_parts(PlaneGeom g, bool half) {
    const int hh = g.rows / 2, lx = inv_lx_for(hh, g.cols, EPI_KFWD_POLY3);
    const int threads = lx * plan_desc_inv(hh).t, nw = (threads + 63) / 64;
    const int tiles = half ? (g.cols / 2) / lx + 1 : g.cols / lx;
    return tiles * nw;
}

template <int HH, int SRC> static void launchA_fwd_t(hipStream_t s, int n_items, AArgs a, size_t min_lds = 0) {
    a.n_items = n_items;
    dim3 grid((a.cols / FCfg<HH>::LX) * n_items), block(FCfg<HH>::NT);
    if (SRC == SRC_POLAR_H || SRC == SRC_POLAR_Q || SRC == SRC_POLAR_T) {
        // tile groups start on a multiple of 8 blocks, so that a block's place inside its group says which XCD it runs on
        // (blocks go to the XCDs round-robin); the few blocks of padding return at once
        const int G = a.polar_group, tiles = a.cols / FCfg<HH>::LX;
        grid = dim3((unsigned)((tiles / G) * ((G * n_items + 7) / 8 * 8)));
    }
    const size_t bytes = std::max(fwd_lds_bytes<HH, SRC>(), min_lds);
    static size_t lds_cap = 65536;                           // largest dynamic LDS size this instantiation may launch with
    if (bytes > lds_cap &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&kA_fwd<HH, SRC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess)
        lds_cap = bytes;
    hipLaunchKernelGGL((kA_fwd<HH, SRC>), grid, block, bytes, s, a);
}
template <int HH, int EPI, int LXO = 0> static void launchA_inv_t(hipStream_t s, int n_items, int nz, AArgs a) {
    if constexpr (LXO == 0 && pref_lx(HH) > 0) {
        if (a.cols % pref_lx(HH) == 0) { launchA_inv_t<HH, EPI, pref_lx(HH)>(s, n_items, nz, a); return; }
    }
    a.n_items = n_items; a.tw_f = a.twI_f; a.tw_i = a.twI_i;       // tables of the inverse-kernel plan
    using C = ICfg<HH, EPI, LXO>;
    const int nbx = a.cols / C::LX;
    // zz_tiles (flag -> tile count): columns [0, min(W/2, zz_tiles - 1)] rounded up to whole tiles
    if (a.zz_tiles > 0) a.zz_tiles = std::min(a.cols / 2, a.zz_tiles - 1) / C::LX + 1;
    dim3 grid(a.zz_tiles > 0 ? (a.zz_tiles + (epi_is_kfwd(EPI) ? nbx : 0)) * n_items : nbx * n_items * nz), block(C::NT);
    static const bool big_lds = (C::BYTES > 65536) &&
        (hipFuncSetAttribute(reinterpret_cast<const void*>(&kA_inv<HH, EPI, LXO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::BYTES) == hipSuccess);
    (void)big_lds;
#ifdef KCC_ABLATE
    static const size_t lds_pad = tune_env("NIK_LDS_PAD_A") ? (size_t)atoi(tune_env("NIK_LDS_PAD_A")) : 0;   // occupancy experiments
    hipLaunchKernelGGL((kA_inv<HH, EPI, LXO>), grid, block, std::min<size_t>(C::BYTES + lds_pad, 65536), s, a);
    return;
#endif
    hipLaunchKernelGGL((kA_inv<HH, EPI, LXO>), grid, block, C::BYTES, s, a);
}

// polar forward kernel for `qs` first-pass points per thread and segment (one of the sizes fwd_geom() offers)
#ifndef KCC_POLAR_GROUP
#define KCC_POLAR_GROUP 10
#endif
template <int HH> static void polar_launch(hipStream_t s, int n_items, const AArgs& a_in, int qs, size_t lds) {
    constexpr int RF = Dir<typename FCfg<HH>::P, false>::RF;
    AArgs a = a_in;
    // tiles per group: the largest divisor of the tile count that is <= the wanted group size ($NIK_POLAR_GROUP; 1 = tile-major)
    static const int want = tune_env("NIK_POLAR_GROUP") ? std::max(1, atoi(tune_env("NIK_POLAR_GROUP"))) : KCC_POLAR_GROUP;
    const int tiles = a.cols / FCfg<HH>::LX;
    int G = std::min(want, tiles);
    while (tiles % G) --G;
    a.polar_group = G;
    if constexpr (RF % 2 == 0) { if (qs == RF / 2) { launchA_fwd_t<HH, SRC_POLAR_H>(s, n_items, a, lds); return; } }
    if constexpr (polar_qs_mid(RF) > 1) { if (qs == polar_qs_mid(RF)) { launchA_fwd_t<HH, SRC_POLAR_T>(s, n_items, a, lds); return; } }
    launchA_fwd_t<HH, SRC_POLAR_Q>(s, n_items, a, lds);
}
static AArgs base_args(PlaneGeom g, Tables t) {
    AArgs a{};
    a.ablate = ablate_flags(); a.rev = g_launch_rev; a.rows = g.rows; a.cols = g.cols; a.hr = g.hr; a.tw_f = t.half_f; a.tw_i = t.half_i; a.tw_full = t.tw_full; a.twI_f = t.halfI_f; a.twI_i = t.halfI_i;
    return a;
}

#define DISPATCH_HALF(h, CALL)            \
    switch (h) {                          \
        case 30:  { CALL(30);  break; }   \
        case 60:  { CALL(60);  break; }   \
        case 120: { CALL(120); break; }   \
        case 224: { CALL(224); break; }   \
        case 240: { CALL(240); break; }   \
        case 256: { CALL(256); break; }   \
        case 360: { CALL(360); break; }   \
        case 384: { CALL(384); break; }   \
        case 600: { CALL(600); break; }   \
        default: break;                   \
    }

void launch_A_fwd_plane(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float* src, size_t src_stride, int src_pitch,
                        const int* src_idx, float2* dst, size_t dst_stride) {
    AArgs a = base_args(g, t);
    a.src = src; a.src_stride = src_stride; a.src_pitch = src_pitch; a.src_idx = src_idx; a.spec = dst; a.spec_stride = dst_stride;
#define CALL(HH) launchA_fwd_t<HH, SRC_PLANE>(s, n_items, a)
    DISPATCH_HALF(g.rows / 2, CALL)
#undef CALL
}
void launch_A_fwd_rot(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float* arena_img, size_t img_stride, int img_pitch,
                      const int* img_slot, const int* rot_tab, const int* rot_index, float2* dst, size_t dst_stride, float* dbg_plane) {
    AArgs a = base_args(g, t);
    a.src = arena_img; a.src_stride = img_stride; a.src_pitch = img_pitch; a.src_idx = img_slot; a.rot_tab = rot_tab; a.rot_index = rot_index;
    a.spec = dst; a.spec_stride = dst_stride; a.dbg_plane = dbg_plane;
#define CALL(HH) launchA_fwd_t<HH, SRC_ROT>(s, n_items, a)
    DISPATCH_HALF(g.rows / 2, CALL)
#undef CALL
}
void launch_A_fwd_polar(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float* S, size_t s_stride,
                        int H, int W, const PolarPlan& pp, float2* dst, size_t dst_stride, float* dbg_plane) {
    AArgs a = base_args(g, t);
    a.src = S; a.src_stride = s_stride; a.H = H; a.W = W; a.SP = H + 2; a.spec = dst; a.spec_stride = dst_stride;
    a.polar_chunks = pp.chunks; a.polar_seg_first = pp.seg_first; a.polar_pts = pp.pts; a.dbg_plane = dbg_plane;
#define CALL(HH) polar_launch<HH>(s, n_items, a, pp.qs, pp.lds_bytes)
    DISPATCH_HALF(g.rows / 2, CALL)
#undef CALL
}
template <int HH> static void launchA_fwd_u8_t(hipStream_t s, int n_items, AArgs a) {
    a.n_items = n_items;
    const int nbx = a.cols / FCfg<HH>::LX;
    static const int tpw_env = [] { const char* e = tune_env("NIK_U8_TPW"); return e ? atoi(e) : 0; }();
    int tpw = tpw_env > 0 ? tpw_env : KCC_U8_TPW;
    while (tpw > 1 && nbx % tpw) --tpw;                      // tiles per workgroup must divide the tiles of an image
    dim3 grid((nbx / tpw) * n_items), block(FCfg<HH>::NT);
    const size_t bytes = FCfg<HH>::BYTES;
    static const bool big_lds = (FCfg<HH>::BYTES > 65536) &&
        (hipFuncSetAttribute(reinterpret_cast<const void*>(&kA_fwd_u8<HH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FCfg<HH>::BYTES) == hipSuccess);
    (void)big_lds;
    hipLaunchKernelGGL((kA_fwd_u8<HH>), grid, block, bytes, s, a, tpw);
}
void launch_A_fwd_u8(hipStream_t s, int n_items, PlaneGeom g, Tables t, const uint8_t* src, size_t src_stride, int src_pitch,
                     uint8_t* keep, size_t keep_stride, int keep_pitch, const int* keep_slot, float2* dst, size_t dst_stride) {
    AArgs a = base_args(g, t);
    a.src8 = src; a.src8_stride = src_stride; a.src8_pitch = src_pitch;
    a.dst8 = keep; a.dst8_stride = keep_stride; a.dst8_pitch = keep_pitch; a.dst8_slot = keep_slot;
    a.spec = dst; a.spec_stride = dst_stride;
#define CALL(HH) launchA_fwd_u8_t<HH>(s, n_items, a)
    DISPATCH_HALF(g.rows / 2, CALL)
#undef CALL
}
void launch_A_fwd_rot8(hipStream_t s, int n_items, PlaneGeom g, Tables t, const uint8_t* arena_u8, size_t img_stride, int img_pitch,
                       const int* img_slot, const int* rot_tab, const int* rot_index, float2* dst, size_t dst_stride, float* dbg_plane) {
    AArgs a = base_args(g, t);
    a.src8 = arena_u8; a.src8_stride = img_stride; a.src8_pitch = img_pitch; a.src_idx = img_slot; a.rot_tab = rot_tab; a.rot_index = rot_index;
    a.spec = dst; a.spec_stride = dst_stride; a.dbg_plane = dbg_plane;
#define CALL(HH) launchA_fwd_t<HH, SRC_ROT8>(s, n_items, a)
    DISPATCH_HALF(g.rows / 2, CALL)
#undef CALL
}
void launch_A_inv_real(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* src, size_t src_stride,
                       float* dst, size_t dst_stride) {
    AArgs a = base_args(g, t);
    a.spec = const_cast<float2*>(src); a.spec_stride = src_stride; a.real_out = dst; a.real_stride = dst_stride;
#define CALL(HH) launchA_inv_t<HH, EPI_REAL>(s, n_items, 1, a)
    DISPATCH_HALF(g.rows / 2, CALL)
#undef CALL
}
void launch_A_inv_shifted(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* src, size_t src_stride,
                          float* S, size_t s_stride, int need_cols, bool fix_zero) {
    AArgs a = base_args(g, t);
    a.spec = const_cast<float2*>(src); a.spec_stride = src_stride; a.real_out = S; a.real_stride = s_stride;
    a.zz_tiles = need_cols > 0 ? need_cols + 1 : 0;          // (the launcher turns "columns <= need" into the tile count)
    a.fix_zero = (fix_zero && need_cols > 0) ? 1 : 0;
#define CALL(HH) launchA_inv_t<HH, EPI_SHIFTED>(s, n_items, 1, a)
    DISPATCH_HALF(g.rows / 2, CALL)
#undef CALL
}
void launch_A_inv_kernel_fwd(hipStream_t s, int n_items, PlaneGeom g, Tables t, float2* buf, size_t item_stride,
                             size_t plane_stride, KernelFn fn, unsigned* maxbuf, const float* energy,
                             int plane_first, int n_planes, bool zz_half) {
    AArgs a = base_args(g, t);
    a.spec = buf; a.spec_stride = item_stride; a.plane_stride = plane_stride; a.fn = fn; a.maxbuf = maxbuf; a.energy = energy;
    a.plane_first = plane_first; a.n_planes = n_planes;
    a.zz_tiles = (zz_half && plane_first == 0 && n_planes == 2) ? g.cols / 2 + 1 : 0;      // (the launcher turns it into the tile count)
    if (fn.type == 1) {
#define CALL(HH) launchA_inv_t<HH, EPI_KFWD_GAUSS>(s, n_items, n_planes, a)
        DISPATCH_HALF(g.rows / 2, CALL)
#undef CALL
    } else if (fn.power == 3) {
#define CALL(HH) launchA_inv_t<HH, EPI_KFWD_POLY3>(s, n_items, n_planes, a)
        DISPATCH_HALF(g.rows / 2, CALL)
#undef CALL
    } else {
#define CALL(HH) launchA_inv_t<HH, EPI_KFWD_POLYN>(s, n_items, n_planes, a)
        DISPATCH_HALF(g.rows / 2, CALL)
#undef CALL
    }
}
void launch_A_inv_argmax_win(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* src, size_t src_stride,
                             Partial* partials, int partial_stride, const int* win_row, const int* win_col, int radius, int mirror) {
    AArgs a = base_args(g, t);
    a.spec = const_cast<float2*>(src); a.spec_stride = src_stride; a.partials = partials; a.partial_stride = partial_stride;
    a.win_row = win_row; a.win_col = win_col; a.win_radius = radius; a.win_mirror = mirror;
#define CALL(HH) launchA_inv_t<HH, EPI_ARGMAX_WIN>(s, n_items, 1, a)
    DISPATCH_HALF(g.rows / 2, CALL)
#undef CALL
}
void launch_A_inv_argmax(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* src, size_t src_stride,
                         Partial* partials, int partial_stride) {
    AArgs a = base_args(g, t);
    a.spec = const_cast<float2*>(src); a.spec_stride = src_stride; a.partials = partials; a.partial_stride = partial_stride;
#define CALL(HH) launchA_inv_t<HH, EPI_ARGMAX>(s, n_items, 1, a)
    DISPATCH_HALF(g.rows / 2, CALL)
#undef CALL
}

// ------------------------------------------------------------------------------------------------
// B-type kernels
// ------------------------------------------------------------------------------------------------
enum { B_FWD = 0, B_FWD_ABS_INV = 1, B_MUL_INV = 2, B_FWD_MUL_INV = 3, B_SOLVE_INV = 4, B_INV = 5,
       // per-keyframe Kzz cache: zz-only / xz-only halves of the above and the solve that reads the cached Kzz
       B_ZZ_INV = 6, B_MUL_INV_X = 7, B_FWD_MUL_INV_X = 8, B_SOLVE_CACHED = 9 };

struct BArgs {
    int cols, hr, ablate, rev;
    CCP tw_f, tw_i;
    CCP twA_f, twA_i;                                             // tables of PlanAlt (single-plane modes)
    CCP src; size_t src_stride; const int* src_idx;     // primary input
    CCP zsrc; size_t z_stride; const int* z_idx;        // Z (key) spectra
    size_t in_plane_stride;                                       // SOLVE: plane 1 offset inside src item
    CP dst; size_t dst_stride; const int* dst_slot;          // primary output
    CP dst2; size_t dst2_stride; const int* dst2_slot;       // secondary output (FWD_MUL_INV*: the forward spectrum X itself)
    size_t out_plane_stride;                                      // MUL_INV: plane 1 offset inside dst item
    const unsigned* maxbuf; int n_parts[2]; float lambda;     // running-max parts of the kernel planes (see kA_inv kernel_fwd)
    int zz_half;                                                  // SOLVE_INV: plane 0 holds only the columns <= N/2 (Hermitian);
                                                                  // (FWD_)MUL_INV: > 0 = number of zz-plane columns to store
    unsigned* maxbuf_zero;                                        // MUL_INV: running-max slots to reset for the next stage
    CCP kzz; size_t kzz_stride; const unsigned* mzz;   // SOLVE_CACHED: per-slot Kzz spectra and max (slot = z_idx[item])
};

// lines per workgroup: fewer for the modes that hold two planes' exchange buffers (LDS-limited occupancy)
#ifndef KCC_BLK_MID
#define KCC_BLK_MID 8
#endif
#ifndef KCC_BLK_MID2
#define KCC_BLK_MID2 5
#endif
#ifndef KCC_BLK_BIG
#define KCC_BLK_BIG 4
#endif
#ifndef KCC_BLK_HUGE
#define KCC_BLK_HUGE 2
#endif
#ifndef KCC_BLK_HUGE2
#define KCC_BLK_HUGE2 1
#endif
#ifndef KCC_BLK_BIG2
#define KCC_BLK_BIG2 3
#endif
template <int N, int MODE> struct BCfg {
    using P0 = PlanFor<N>;
    static constexpr int T0 = P0::T;
    // exchange buffers per line: two for the modes that transform two planes at once, else one
    // SEQ: the two planes of the mode go through ONE exchange buffer one after the other -- half the LDS, twice the
    // waves per CU, one more barrier per chain.  Pays for the ridge solve of the mid-size lines (measured at 480:
    // 0.255 -> 0.220 ms; loses for the 640-point lines and for the product kernels, which stay two-buffer).
#ifndef KCC_B_SEQ_SOLVE
#define KCC_B_SEQ_SOLVE 1
#endif
#ifndef KCC_B_SEQ_TMAX
#define KCC_B_SEQ_TMAX 64
#endif
    // KCC_SOLVE_ALT (tuning, round 6): the ridge solve of the long lines (PlanFor's T >= 128: 1280 points) on PlanAlt in the
    // single-buffer form -- with -DKCC_PA1280=32,40 a two-pass plan of 40 threads per line instead of 20 x 8 x 8 on 160
#ifndef KCC_SOLVE_ALT
#define KCC_SOLVE_ALT 0
#endif
    static constexpr bool SOLVE_ALT = KCC_SOLVE_ALT && MODE == 4 && T0 >= 128;
    static constexpr bool SEQ = KCC_B_SEQ_SOLVE && MODE == 4 && (T0 < KCC_B_SEQ_TMAX || SOLVE_ALT);
    static constexpr int NV = (!SEQ && (MODE == 2 || MODE == 3 || MODE == 4)) ? 2 : 1;
    // single-plane modes (not the single-buffer solve, which still holds two planes in registers) use PlanAlt
    static constexpr bool ALT = !(MODE == 2 || MODE == 3 || MODE == 4) || SOLVE_ALT;
    using P = typename std::conditional<ALT, PlanAlt<N>, PlanFor<N>>::type;
    static constexpr int T = P::T;
    static constexpr int LK = (T >= 128) ? (NV == 2 ? KCC_BLK_HUGE2 : KCC_BLK_HUGE) : (T >= 64 ? (NV == 2 ? KCC_BLK_BIG2 : KCC_BLK_BIG)
                                                        : (T >= 20 ? (NV == 2 ? KCC_BLK_MID2 : KCC_BLK_MID) : 16));
    static constexpr int NT = LK * T;
    static constexpr int EPITCH = ((P::EXT + 31 - (T % 32)) / 32) * 32 + (T % 32);
    static constexpr size_t BYTES = (size_t)NV * LK * EPITCH * sizeof(cf2);
};

template <int RR>
__device__ __forceinline__ void load_strided(cf2 (&v)[RR], const cf2* __restrict__ p, int stride, bool ok) {
    if (ok) {
#pragma unroll
        for (int q = 0; q < RR; ++q) v[q] = p[q * stride];
    } else {
        zero_fill(v);
    }
}
template <int RR>
__device__ __forceinline__ void store_strided(const cf2 (&v)[RR], cf2* __restrict__ p, int stride) {
#pragma unroll
    for (int q = 0; q < RR; ++q) p[q * stride] = v[q];
}

// max over the running-max parts of one kernel plane; called by the (full) first wave of a workgroup, result in every lane
__device__ __forceinline__ float parts_max(const unsigned* __restrict__ p, int n, int lane) {
    unsigned m = 0u;                                         // non-negative floats order as their bit patterns
    for (int i = lane; i < n; i += 64) m = max(m, p[i]);
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    return __uint_as_float(m);
}

template <int N, int MODE>
__global__ __launch_bounds__((BCfg<N, MODE>::NT)) void kB(BArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = BCfg<N, MODE>; using P = typename C::P; using DF = Dir<P, false>; using DI = Dir<P, true>;
    static_assert(DF::RL == DI::RF && DF::ML == DI::MF && DI::RL == DF::RF, "direction layouts must chain");
    cf2* lds = reinterpret_cast<cf2*>(smem);
    __shared__ float s_rmax[2];
    if (ABL(a, 8)) return;
    const unsigned tid = threadIdx.x, lk = tid / (unsigned)C::T, j = tid - lk * C::T;
    const int item = a.rev ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y, k = blockIdx.x * C::LK + (int)lk;      // spectrum line (row index of the half spectrum)
    const bool valid0 = k < a.hr;
    const bool valid = valid0 && !ABL(a, 1);          // loads
    const bool vst = valid0 && !ABL(a, 2);            // stores
    const bool nofft = ABL(a, 4);
    const size_t loff = (size_t)k * N + j;
    constexpr int NVM = C::NV;
    // single-plane modes whose plan puts a line inside one wave (PlanAlt<640>: 32 threads per line): no workgroup barrier
    // between the passes of a chain, the waves of a workgroup run their lines independently
    constexpr bool WLB = KCC_WAVE_LOCAL && C::ALT && (64 % C::T == 0);
    cf2* const ex1[1] = { lds + (NVM * lk) * C::EPITCH };
    cf2* const ex2[2] = { lds + (NVM * lk) * C::EPITCH, lds + (NVM * lk + NVM - 1) * C::EPITCH };

    if (MODE == B_FWD || MODE == B_INV) {
        constexpr bool INV = (MODE == B_INV);
        using D = Dir<P, INV>;
        cf2 vin[1][D::RF], vout[1][D::RL];
        load_strided(vin[0], a.src + (size_t)(a.src_idx ? a.src_idx[item] : item) * a.src_stride + loff, D::MF, valid && j < D::MF);
        if (!nofft) fft_chain<P, INV, 1, WLB>(vin, vout, j, ex1, INV ? a.tw_i : a.tw_f);
        if (vst && j < D::ML)
            store_strided(vout[0], a.dst + (size_t)(a.dst_slot ? a.dst_slot[item] : item) * a.dst_stride + loff, D::ML);
    } else if (MODE == B_FWD_ABS_INV) {
        // fft_result = FFT(image);  IFFT(fft_result.abs())   (correlation_flow.cc:91-92)
        cf2 vin[1][DF::RF], f[1][DF::RL], o[1][DI::RL];
        load_strided(vin[0], a.src + (size_t)item * a.src_stride + loff, DF::MF, valid && j < DF::MF);
        if (!nofft) fft_chain<P, false, 1, WLB>(vin, f, j, ex1, a.tw_f);
        if (vst && j < DF::ML)
            store_strided(f[0], a.dst + (size_t)(a.dst_slot ? a.dst_slot[item] : item) * a.dst_stride + loff, DF::ML);
#pragma unroll
        for (int q = 0; q < DF::RL; ++q) f[0][q] = mk2(sqrtf(f[0][q].x * f[0][q].x + f[0][q].y * f[0][q].y), 0.f);
        line_sync<WLB>();
        if (!nofft) fft_chain<P, true, 1, WLB>(f, o, j, ex1, a.tw_i);
        if (vst && j < DI::ML) {
            cf2* d2 = a.dst2 + (size_t)item * a.dst2_stride + loff;
            if (a.zz_half > 0) {                              // only the columns the even-half inverse row pass reads
#pragma unroll
                for (int q = 0; q < DI::RL; ++q) if ((int)j + q * DI::ML < a.zz_half) d2[q * DI::ML] = o[0][q];
            } else {
                store_strided(o[0], d2, DI::ML);
            }
        }
    } else if (MODE == B_MUL_INV || MODE == B_FWD_MUL_INV) {
        // xzf = xf * zf.conjugate() for (z,z) and (x,z)   (correlation_flow.cc:210-211,220-221)
        cf2 pr[2][DI::RF], o[2][DI::RL], zv[DI::RF];
        // the key spectrum line is needed only after the forward chain: issue its loads first (latency hidden)
        load_strided(zv, a.zsrc + (size_t)(a.z_idx ? a.z_idx[item] : item) * a.z_stride + loff, DI::MF, valid && j < DI::MF);
        if (MODE == B_FWD_MUL_INV) {
            cf2 vin[1][DF::RF], x[1][DF::RL];
            load_strided(vin[0], a.src + (size_t)(a.src_idx ? a.src_idx[item] : item) * a.src_stride + loff, DF::MF, valid && j < DF::MF);
            if (!nofft) fft_chain<P, false, 1, WLB>(vin, x, j, ex1, a.tw_f);
            if (a.dst2 && vst && j < DF::ML)                 // X is a result of its own (the frame's spectrum): keep it
                store_strided(x[0], a.dst2 + (size_t)(a.dst2_slot ? a.dst2_slot[item] : item) * a.dst2_stride + loff, DF::ML);
#pragma unroll
            for (int q = 0; q < DI::RF; ++q) pr[1][q] = cmulc(x[0][q], zv[q]);
            __syncthreads();
        } else {
            cf2 xv[DI::RF];
            load_strided(xv, a.src + (size_t)(a.src_idx ? a.src_idx[item] : item) * a.src_stride + loff, DI::MF, valid && j < DI::MF);
#pragma unroll
            for (int q = 0; q < DI::RF; ++q) pr[1][q] = cmulc(xv[q], zv[q]);
        }
#pragma unroll
        for (int q = 0; q < DI::RF; ++q) pr[0][q] = mk2(zv[q].x * zv[q].x + zv[q].y * zv[q].y, 0.f);
        if (!nofft) {
            if (C::SEQ) {
                cf2 (&p0)[1][DI::RF] = reinterpret_cast<cf2 (&)[1][DI::RF]>(pr[0]); cf2 (&p1)[1][DI::RF] = reinterpret_cast<cf2 (&)[1][DI::RF]>(pr[1]);
                cf2 (&o0)[1][DI::RL] = reinterpret_cast<cf2 (&)[1][DI::RL]>(o[0]);  cf2 (&o1)[1][DI::RL] = reinterpret_cast<cf2 (&)[1][DI::RL]>(o[1]);
                fft_chain<P, true, 1, WLB>(p0, o0, j, ex1, a.tw_i);
                __syncthreads();
                fft_chain<P, true, 1, WLB>(p1, o1, j, ex1, a.tw_i);
            } else {
                fft_chain<P, true, 2>(pr, o, j, ex2, a.tw_i);
            }
        }
        if (vst && j < DI::ML) {
            cf2* d = a.dst + (size_t)item * a.dst_stride + loff;
            if (a.zz_half > 0) {
                // only the columns the Hermitian-half kernel_fwd reads (x < zz_half) of the zz plane are kept
#pragma unroll
                for (int q = 0; q < DI::RL; ++q) if ((int)j + q * DI::ML < a.zz_half) d[q * DI::ML] = o[0][q];
            } else {
                store_strided(o[0], d, DI::ML);
            }
            store_strided(o[1], d + a.out_plane_stride, DI::ML);
        }
    } else if (MODE == B_ZZ_INV) {
        // Kzz half of the kernel stage: |Z|^2 -> inverse col FFT (plane 0)
        cf2 zv[1][DI::RF], o[1][DI::RL];
        load_strided(zv[0], a.zsrc + (size_t)(a.z_idx ? a.z_idx[item] : item) * a.z_stride + loff, DI::MF, valid && j < DI::MF);
#pragma unroll
        for (int q = 0; q < DI::RF; ++q) zv[0][q] = mk2(zv[0][q].x * zv[0][q].x + zv[0][q].y * zv[0][q].y, 0.f);
        if (!nofft) fft_chain<P, true, 1, WLB>(zv, o, j, ex1, a.tw_i);
        if (vst && j < DI::ML) store_strided(o[0], a.dst + (size_t)item * a.dst_stride + loff, DI::ML);
    } else if (MODE == B_MUL_INV_X || MODE == B_FWD_MUL_INV_X) {
        // Kxz half: X conj Z -> inverse col FFT (plane 1)
        cf2 pr[1][DI::RF], o[1][DI::RL], zv[DI::RF];
        load_strided(zv, a.zsrc + (size_t)(a.z_idx ? a.z_idx[item] : item) * a.z_stride + loff, DI::MF, valid && j < DI::MF);
        if (MODE == B_FWD_MUL_INV_X) {
            cf2 vin[1][DF::RF], x[1][DF::RL];
            load_strided(vin[0], a.src + (size_t)(a.src_idx ? a.src_idx[item] : item) * a.src_stride + loff, DF::MF, valid && j < DF::MF);
            if (!nofft) fft_chain<P, false, 1, WLB>(vin, x, j, ex1, a.tw_f);
            if (a.dst2 && vst && j < DF::ML)
                store_strided(x[0], a.dst2 + (size_t)(a.dst2_slot ? a.dst2_slot[item] : item) * a.dst2_stride + loff, DF::ML);
#pragma unroll
            for (int q = 0; q < DI::RF; ++q) pr[0][q] = cmulc(x[0][q], zv[q]);
            __syncthreads();
        } else {
            cf2 xv[DI::RF];
            load_strided(xv, a.src + (size_t)(a.src_idx ? a.src_idx[item] : item) * a.src_stride + loff, DI::MF, valid && j < DI::MF);
#pragma unroll
            for (int q = 0; q < DI::RF; ++q) pr[0][q] = cmulc(xv[q], zv[q]);
        }
        if (!nofft) fft_chain<P, true, 1, WLB>(pr, o, j, ex1, a.tw_i);
        if (vst && j < DI::ML) store_strided(o[0], a.dst + (size_t)item * a.dst_stride + a.out_plane_stride + loff, DI::ML);
    } else if (MODE == B_SOLVE_CACHED) {
        // as SOLVE_INV, but Kzz (already transformed) and its max come from the key slot's cache
        cf2 vin[1][DF::RF], kx[1][DF::RL], kz[DF::RL], g[1][DI::RF], o[1][DI::RL];
        const int zslot = a.z_idx[item];
        load_strided(vin[0], a.src + (size_t)item * a.src_stride + a.in_plane_stride + loff, DF::MF, valid && j < DF::MF);
        load_strided(kz, a.kzz + (size_t)zslot * a.kzz_stride + loff, DF::ML, valid && j < DF::ML);
        static_assert(C::NT >= 64, "parts_max needs one full wave");
        if (tid < 64) { const float m = parts_max(a.maxbuf + (size_t)(2 * item + 1) * KCC_MAXPARTS, a.n_parts[1], (int)tid); if (tid == 0) s_rmax[1] = m; }
        if (!nofft) fft_chain<P, false, 1, WLB>(vin, kx, j, ex1, a.tw_f);
        if (WLB) __syncthreads();                                     // (wave-local chains hold no workgroup barrier: s_rmax needs one)
        const float rzz = 1.0f / __uint_as_float(a.mzz[zslot]);      // (s_rmax: written before the barriers inside the chain)
        const float rxz = 1.0f / s_rmax[1];
        const float sg = ((k + (int)j) & 1) ? -1.f : 1.f;
#pragma unroll
        for (int q = 0; q < DF::RL; ++q) {
            const cf2 den = mk2(kz[q].x * rzz + a.lambda, kz[q].y * rzz);
            const cf2 num = mk2(kx[0][q].x * rxz, kx[0][q].y * rxz);
            const float inv = ((DF::ML % 2 == 0 || q % 2 == 0) ? sg : -sg) / (den.x * den.x + den.y * den.y);      // IEEE division, as the reference divides (correlation_flow.cc:171)
            const cf2 gg = cmulc(num, den);
            g[0][q] = mk2(gg.x * inv, gg.y * inv);
        }
        if (!(valid0 && j < DF::ML)) zero_fill(g[0]);
        __syncthreads();
        if (!nofft) fft_chain<P, true, 1, WLB>(g, o, j, ex1, a.tw_i);
        if (vst && j < DI::ML) store_strided(o[0], a.dst + (size_t)item * a.dst_stride + loff, DI::ML);
    } else {
        // H = T/(Kzz + lambda); G = H * Kxz   (correlation_flow.cc:171-172), T[k][l] = (-1)^(k+l)
        cf2 vin[2][DF::RF], kk[2][DF::RL], g[1][DI::RF], o[1][DI::RL];
        const cf2* src = a.src + (size_t)item * a.src_stride + loff;
        if (a.zz_half) {
            // plane 0 (the Kzz kernel plane after its row pass) is Hermitian along x: element x > N/2 = conj(element N - x)
            if (valid && j < DF::MF) {
                const cf2* row = a.src + (size_t)item * a.src_stride + (size_t)k * N;
#pragma unroll
                for (int q = 0; q < DF::RF; ++q) {
                    const int x = (int)j + q * DF::MF;
                    const cf2 v = row[x <= N / 2 ? x : N - x];
                    vin[0][q] = mk2(v.x, x <= N / 2 ? v.y : -v.y);
                }
            } else {
                zero_fill(vin[0]);
            }
        } else {
            load_strided(vin[0], src, DF::MF, valid && j < DF::MF);
        }
        load_strided(vin[1], src + a.in_plane_stride, DF::MF, valid && j < DF::MF);
        static_assert(C::NT >= 64, "parts_max needs one full wave");
        if (tid < 64) {
            const float m0 = parts_max(a.maxbuf + (size_t)(2 * item + 0) * KCC_MAXPARTS, a.n_parts[0], (int)tid);
            const float m1 = parts_max(a.maxbuf + (size_t)(2 * item + 1) * KCC_MAXPARTS, a.n_parts[1], (int)tid);
            if (tid == 0) { s_rmax[0] = m0; s_rmax[1] = m1; }
        }
        if (!nofft) {
            if (C::SEQ) {
                cf2 (&v0)[1][DF::RF] = reinterpret_cast<cf2 (&)[1][DF::RF]>(vin[0]); cf2 (&v1)[1][DF::RF] = reinterpret_cast<cf2 (&)[1][DF::RF]>(vin[1]);
                cf2 (&k0)[1][DF::RL] = reinterpret_cast<cf2 (&)[1][DF::RL]>(kk[0]);  cf2 (&k1)[1][DF::RL] = reinterpret_cast<cf2 (&)[1][DF::RL]>(kk[1]);
#ifdef KCC_UB_NOZZFWD
                (void)v0;
#pragma unroll
                for (int q = 0; q < DF::RL; ++q) k0[0][q] = vin[0][q % DF::RF];
                fft_chain<P, false, 1, WLB>(v1, k1, j, ex1, a.tw_f);
#else
                fft_chain<P, false, 1, WLB>(v0, k0, j, ex1, a.tw_f);
                __syncthreads();
                fft_chain<P, false, 1, WLB>(v1, k1, j, ex1, a.tw_f);
#endif
            } else {
#ifdef KCC_UB_NOZZFWD
                cf2 (&v1)[1][DF::RF] = reinterpret_cast<cf2 (&)[1][DF::RF]>(vin[1]); cf2 (&k1)[1][DF::RL] = reinterpret_cast<cf2 (&)[1][DF::RL]>(kk[1]);
#pragma unroll
                for (int q = 0; q < DF::RL; ++q) kk[0][q] = vin[0][q % DF::RF];
                cf2* const exu[1] = { ex2[1] };
                fft_chain<P, false, 1>(v1, k1, j, exu, a.tw_f);
#else
                fft_chain<P, false, 2>(vin, kk, j, ex2, a.tw_f);
#endif
            }
        }
        const float rzz = 1.0f / s_rmax[0], rxz = 1.0f / s_rmax[1];   // (wave 0 wrote them before the barriers inside the chains)
        // T[k][l] = (-1)^(k + l), l = j + q ML: one sign per thread when ML is even, alternating with q when it is odd (16 x 47 lines)
        const float sg = ((k + (int)j) & 1) ? -1.f : 1.f;
#pragma unroll
        for (int q = 0; q < DF::RL; ++q) {
            const cf2 den = mk2(kk[0][q].x * rzz + a.lambda, kk[0][q].y * rzz);
            const cf2 num = mk2(kk[1][q].x * rxz, kk[1][q].y * rxz);
            const float inv = ((DF::ML % 2 == 0 || q % 2 == 0) ? sg : -sg) / (den.x * den.x + den.y * den.y);      // IEEE division, as the reference divides (correlation_flow.cc:171)
            const cf2 gg = cmulc(num, den);
            g[0][q] = mk2(gg.x * inv, gg.y * inv);
        }
        if (!(valid0 && j < DF::ML)) zero_fill(g[0]);
        __syncthreads();
        if (!nofft) fft_chain<P, true, 1, WLB>(g, o, j, ex1, a.tw_i);
        if (vst && j < DI::ML) store_strided(o[0], a.dst + (size_t)item * a.dst_stride + loff, DI::ML);
    }
}

// (the ring form is a measured no-go for speed -- DESIGN 4.5 -- and is compiled into the TUNING library only: -DKCC_ABLATE)
#ifdef KCC_ABLATE
// ------------------------------------------------------------------------------------------------
// B-type kernels, ring form (round 5): persistent workgroups fed by a loader wave through LDS-DMA
// ------------------------------------------------------------------------------------------------
// The kB kernels above load a tile's spectrum lines into registers (8 bytes per lane) at the start of a workgroup's life and
// nothing is in flight while the workgroup transforms: memory time and arithmetic time ADD.  Every attempt to
// prefetch the next tile paid in VGPRs -- hence in waves.  LDS-DMA costs none: a persistent workgroup is LK lines of
// consumer threads plus ONE loader wave that requests the NEXT tile's lines with global_load_lds_dwordx4 (16 bytes per lane,
// a line is contiguous in the k-major spectrum: the ideal DMA shape) while the consumers transform the current one.  LDS
// holds two SETS of plane buffers; a set receives a tile's input lines in natural order, the consumers pick their first-pass
// points out of it (ds_read_b64, consecutive lanes: conflict-free), and from then on the same set serves as the tile's
// exchange buffers -- the staging costs no LDS beyond the second set.  The loader wave owns the only vmcnt that counts DMAs
// (a wave's loads return in order: a consumer that issued DMAs would wait for them at its next twiddle load), executes the
// consumers' barriers with them (K per tile, counted at compile time) and does its s_waitcnt vmcnt(0) right before the tile's
// first barrier.  Same arithmetic, same order of operations, same results as kB (the parity tests run on both).
#ifndef KCC_RING_AUX
#define KCC_RING_AUX 0              // cache policy of the DMA loads: 0 default, 2 = nt
#endif
#ifndef KCC_RING_LK480
#define KCC_RING_LK480 4
#endif
#ifndef KCC_RING_LK640
#define KCC_RING_LK640 3
#endif
#ifndef KCC_RING_LK640A
#define KCC_RING_LK640A 6           // single-plane kernels (PlanAlt<640>: 32 threads per line)
#endif
#ifndef KCC_RING_LK1280
#define KCC_RING_LK1280 1
#endif
__host__ __device__ constexpr bool ring_two_plane(int mode) { return mode == B_MUL_INV || mode == B_FWD_MUL_INV || mode == B_SOLVE_INV; }
__host__ __device__ constexpr bool ring_mode_ok(int mode) { return ring_two_plane(mode) || mode == B_FWD_ABS_INV; }
__host__ __device__ constexpr int ring_lk(int n, int mode) {
    return n == 480 ? KCC_RING_LK480 : n == 640 ? (ring_two_plane(mode) ? KCC_RING_LK640 : KCC_RING_LK640A) : n == 1280 ? KCC_RING_LK1280 : 0;
}
template <int N, int MODE> struct RCfg {
    static_assert(!PlanFor<N>::PRIME, "ring form: the loader wave mirrors the barrier count of the 2- / 3-pass chains only (fft_chain_prime runs 3 line_syncs)");
    static constexpr bool ALT = !ring_two_plane(MODE);
    using P = typename std::conditional<ALT, PlanAlt<N>, PlanFor<N>>::type;
    static constexpr int T = P::T;
    static constexpr int LK = ring_lk(N, MODE) > 0 ? ring_lk(N, MODE) : 1;
    static constexpr int NTC = LK * T;                               // consumer threads
    static constexpr int NTCW = (NTC + 63) / 64 * 64;                // ... rounded up to whole waves
    static constexpr int NT = NTCW + 64;                             // + the loader wave (the last one)
    static constexpr int EPITCH = ((P::EXT + 31 - (T % 32)) / 32) * 32 + (T % 32);
    static constexpr int NBUF = ALT ? 1 : 2;                         // plane buffers per set
    static constexpr int BUF = (LK * EPITCH + 1) / 2 * 2;            // cf2 per plane buffer (16-byte multiple)
    // both directions' pass-twiddle tables live in LDS (copied once per workgroup): a consumer wave that loaded them from
    // global memory would wait for the stores of its previous tile at the first twiddle it needs (loads and stores share one
    // vmcnt on gfx950 and the compiler must assume they complete out of order: s_waitcnt vmcnt(0))
    static constexpr int TWF = P::NP == 3 ? P::R2 * P::R1 + N : N, TWI = P::NP == 3 ? P::R2 * P::R3 + N : N;
    static constexpr size_t TW_OFF = (size_t)2 * NBUF * BUF * sizeof(cf2);
    static constexpr size_t RMAX_OFF = TW_OFF + (size_t)(TWF + TWI) * sizeof(cf2);
    static constexpr size_t BYTES = RMAX_OFF + 16;                   // ... + the running maxima of the solve, double-buffered
    static constexpr bool WLB = KCC_WAVE_LOCAL && ALT && (64 % T == 0);
    static constexpr int CB = WLB ? 0 : (P::NP == 3 ? 3 : 1);        // workgroup barriers inside one fft_chain
    // barriers a consumer executes per tile (the loader wave joins every one of them)
    static constexpr int K = MODE == B_MUL_INV ? 2 + CB : MODE == B_FWD_ABS_INV ? 2 + 2 * CB + (WLB ? 0 : 1) : 3 + 2 * CB;
    // 16-byte pieces per line of each input plane
    static constexpr int PPR_FULL = N / 2, PPR_HALF = (N / 2 + 2) / 2;
    static constexpr int WGPC = (int)(163840 / BYTES) < 1 ? 1 : (int)(163840 / BYTES);   // workgroups per CU by LDS
    static constexpr int WPS = (WGPC * (NT / 64) + 3) / 4 > 8 ? 8 : (WGPC * (NT / 64) + 3) / 4;   // waves per SIMD that occupancy needs
};

// LDS-DMA of NROWS lines (PPR 16-byte pieces each, row_stride bytes apart in global memory) into lds_dst, piece i of the tile
// at byte 16 i; lines >= nvalid repeat line nvalid - 1 (their results are never stored).  One wave.
template <int PPR, int NROWS>
__device__ __forceinline__ void ring_dma(const char* __restrict__ g0, size_t row_stride, int nvalid, char* lds_dst, int lane) {
    constexpr int TOTAL = PPR * NROWS, ITERS = (TOTAL + 63) / 64;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = it * 64 + lane;
        if (i < TOTAL) {
            int row = i / PPR; const int p = i - row * PPR;
            row = min(row, nvalid - 1);
            __builtin_amdgcn_global_load_lds((glb_void*)(g0 + (size_t)row * row_stride + (size_t)p * 16), (lds_void*)(lds_dst + it * 1024), 16, 0, KCC_RING_AUX);
        }
    }
}
// (ablation builds only: a skipped chain must still execute its barriers -- the loader wave counts them)
template <int CB> __device__ __forceinline__ void chain_barriers_only() {
#pragma unroll
    for (int b = 0; b < CB; ++b) __syncthreads();
}
template <int RR>
__device__ __forceinline__ void load_lds(cf2 (&v)[RR], const cf2* p, int stride, bool ok) {
    if (ok) {
#pragma unroll
        for (int q = 0; q < RR; ++q) v[q] = p[q * stride];
    } else {
        zero_fill(v);
    }
}

template <int N, int MODE>
__global__ __launch_bounds__((RCfg<N, MODE>::NT), (RCfg<N, MODE>::WPS)) void kBr(BArgs a, int n_items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = RCfg<N, MODE>; using P = typename C::P; using DF = Dir<P, false>; using DI = Dir<P, true>;
    static_assert(DF::RL == DI::RF && DF::ML == DI::MF && DI::RL == DF::RF, "direction layouts must chain");
    constexpr int LK = C::LK, SET = C::NBUF * C::BUF;
    cf2* lds = reinterpret_cast<cf2*>(smem);
    float* s_rmax = reinterpret_cast<float*>(smem + C::RMAX_OFF);    // [set][2]
    cf2* const twf = reinterpret_cast<cf2*>(smem + C::TW_OFF), * const twi = twf + C::TWF;
    const int tpi = (a.hr + LK - 1) / LK;                             // tiles per item
    const int total = tpi * n_items;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool noload = ABL(a, 1), nost = ABL(a, 2), nofft = ABL(a, 4);

    if (wave == C::NTCW / 64) {
        // ---------------------------------------------------------------- loader wave
        const int lane = (int)threadIdx.x & 63;
        auto issue = [&](int t, int set) {
            int item = t / tpi; const int k0 = (t - item * tpi) * LK;
            if (a.rev) item = n_items - 1 - item;
            const int nvalid = min(LK, a.hr - k0);
            char* dst = smem + (size_t)set * SET * sizeof(cf2);
            const size_t row = (size_t)N * sizeof(cf2), off = (size_t)k0 * row;
            if (noload) return;
            if (MODE == B_SOLVE_INV) {
                const char* src = reinterpret_cast<const char*>(a.src.p + (size_t)item * a.src_stride) + off;
                if (a.zz_half) ring_dma<C::PPR_HALF, LK>(src, row, nvalid, dst, lane);
                else ring_dma<C::PPR_FULL, LK>(src, row, nvalid, dst, lane);
                ring_dma<C::PPR_FULL, LK>(src + a.in_plane_stride * sizeof(cf2), row, nvalid, dst + C::BUF * sizeof(cf2), lane);
                // the running maxima of the item's two kernel planes (kA_inv kernel_fwd filed them as parts): folded HERE, by the
                // wave whose vmcnt holds only loads, and handed over through LDS with the tile
                const float m0 = parts_max(a.maxbuf + (size_t)(2 * item + 0) * KCC_MAXPARTS, a.n_parts[0], lane);
                const float m1 = parts_max(a.maxbuf + (size_t)(2 * item + 1) * KCC_MAXPARTS, a.n_parts[1], lane);
                if (lane == 0) { s_rmax[2 * set] = m0; s_rmax[2 * set + 1] = m1; }
            } else if (MODE == B_FWD_ABS_INV) {
                ring_dma<C::PPR_FULL, LK>(reinterpret_cast<const char*>(a.src.p + (size_t)item * a.src_stride) + off, row, nvalid, dst, lane);
            } else {
                const int xi = a.src_idx ? a.src_idx[item] : item, zi = a.z_idx ? a.z_idx[item] : item;
                ring_dma<C::PPR_FULL, LK>(reinterpret_cast<const char*>(a.src.p + (size_t)xi * a.src_stride) + off, row, nvalid, dst, lane);
                ring_dma<C::PPR_FULL, LK>(reinterpret_cast<const char*>(a.zsrc.p + (size_t)zi * a.z_stride) + off, row, nvalid, dst + C::BUF * sizeof(cf2), lane);
            }
        };
        int t = (int)blockIdx.x, n = 0;
        if (t < total) issue(t, 0);
        for (; t < total; t += (int)gridDim.x, ++n) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // tile n has landed ...
            asm volatile("s_barrier" ::: "memory");                   // ... and the consumers are done with the other set
            if (t + (int)gridDim.x < total) issue(t + (int)gridDim.x, (n + 1) & 1);
#pragma unroll
            for (int b = 1; b < C::K; ++b) asm volatile("s_barrier" ::: "memory");
        }
        return;
    }

    // -------------------------------------------------------------------- consumer waves
    const bool live = threadIdx.x < (unsigned)C::NTC;                 // (the last consumer wave may be partly idle)
    const unsigned tid = threadIdx.x, lk = live ? tid / (unsigned)C::T : 0u, j = live ? tid - lk * C::T : (unsigned)C::T;
    constexpr bool WLB = C::WLB;
    for (int i = (int)tid; i < C::TWF; i += C::NTCW) twf[i] = a.tw_f[i];
    for (int i = (int)tid; i < C::TWI; i += C::NTCW) twi[i] = a.tw_i[i];
    int n = 0;
    for (int t = (int)blockIdx.x; t < total; t += (int)gridDim.x, ++n) {
        int item = t / tpi; const int k0 = (t - item * tpi) * LK;
        if (a.rev) item = n_items - 1 - item;
        const int k = k0 + (int)lk;
        const bool valid0 = live && k < a.hr, vst = valid0 && !nost;
        const size_t loff = (size_t)k * N + j;
        cf2* const set = lds + (size_t)(n & 1) * SET;
        cf2* const b0 = set, * const b1 = set + (C::NBUF - 1) * C::BUF;
        cf2* const ex1[1] = { b0 + lk * C::EPITCH };
        cf2* const ex2[2] = { b0 + lk * C::EPITCH, b1 + lk * C::EPITCH };
        __syncthreads();                                              // B0: the tile's lines are in LDS
        if (MODE == B_FWD_ABS_INV) {
            cf2 vin[1][DF::RF], f[1][DF::RL], o[1][DI::RL];
            load_lds(vin[0], b0 + lk * N + j, DF::MF, j < DF::MF);
            __syncthreads();                                          // B1: staged lines consumed before the exchange overwrites them
            if (!nofft) fft_chain<P, false, 1, WLB>(vin, f, j, ex1, twf); else chain_barriers_only<C::CB>();
            if (vst && j < DF::ML)
                store_strided(f[0], a.dst + (size_t)(a.dst_slot ? a.dst_slot[item] : item) * a.dst_stride + loff, DF::ML);
#pragma unroll
            for (int q = 0; q < DF::RL; ++q) f[0][q] = mk2(sqrtf(f[0][q].x * f[0][q].x + f[0][q].y * f[0][q].y), 0.f);
            line_sync<WLB>();
            if (!nofft) fft_chain<P, true, 1, WLB>(f, o, j, ex1, twi); else chain_barriers_only<C::CB>();
            if (vst && j < DI::ML) {
                cf2* d2 = a.dst2 + (size_t)item * a.dst2_stride + loff;
                if (a.zz_half > 0) {
#pragma unroll
                    for (int q = 0; q < DI::RL; ++q) if ((int)j + q * DI::ML < a.zz_half) d2[q * DI::ML] = o[0][q];
                } else {
                    store_strided(o[0], d2, DI::ML);
                }
            }
        } else if (MODE == B_MUL_INV || MODE == B_FWD_MUL_INV) {
            cf2 pr[2][DI::RF], o[2][DI::RL];
            if (MODE == B_FWD_MUL_INV) {
                cf2 vin[1][DF::RF], x[1][DF::RL];
                load_lds(vin[0], b0 + lk * N + j, DF::MF, j < DF::MF);
                __syncthreads();                                      // B1
                if (!nofft) fft_chain<P, false, 1, WLB>(vin, x, j, ex1, twf); else chain_barriers_only<C::CB>();
                if (a.dst2 && vst && j < DF::ML)
                    store_strided(x[0], a.dst2 + (size_t)(a.dst2_slot ? a.dst2_slot[item] : item) * a.dst2_stride + loff, DF::ML);
                // the key line is picked up only now (plane buffer 1 is untouched by the single-plane chain): no registers held across it
                load_lds(pr[0], b1 + lk * N + j, DI::MF, j < DI::MF);
#pragma unroll
                for (int q = 0; q < DI::RF; ++q) pr[1][q] = cmulc(x[0][q], pr[0][q]);
            } else {
                load_lds(pr[1], b0 + lk * N + j, DI::MF, j < DI::MF);
                load_lds(pr[0], b1 + lk * N + j, DI::MF, j < DI::MF);
#pragma unroll
                for (int q = 0; q < DI::RF; ++q) pr[1][q] = cmulc(pr[1][q], pr[0][q]);
            }
#pragma unroll
            for (int q = 0; q < DI::RF; ++q) pr[0][q] = mk2(pr[0][q].x * pr[0][q].x + pr[0][q].y * pr[0][q].y, 0.f);
            __syncthreads();                                          // B2
            if (!nofft) fft_chain<P, true, 2>(pr, o, j, ex2, twi); else chain_barriers_only<C::CB>();
            if (vst && j < DI::ML) {
                cf2* d = a.dst + (size_t)item * a.dst_stride + loff;
                if (a.zz_half > 0) {
#pragma unroll
                    for (int q = 0; q < DI::RL; ++q) if ((int)j + q * DI::ML < a.zz_half) d[q * DI::ML] = o[0][q];
                } else {
                    store_strided(o[0], d, DI::ML);
                }
                store_strided(o[1], d + a.out_plane_stride, DI::ML);
            }
        } else {
            // B_SOLVE_INV
            cf2 vin[2][DF::RF], kk[2][DF::RL], g[1][DI::RF], o[1][DI::RL];
            if (a.zz_half) {
                if (j < DF::MF) {
                    const cf2* row = b0 + lk * (2 * C::PPR_HALF);
#pragma unroll
                    for (int q = 0; q < DF::RF; ++q) {
                        const int x = (int)j + q * DF::MF;
                        const cf2 v = row[x <= N / 2 ? x : N - x];
                        vin[0][q] = mk2(v.x, x <= N / 2 ? v.y : -v.y);
                    }
                } else {
                    zero_fill(vin[0]);
                }
            } else {
                load_lds(vin[0], b0 + lk * N + j, DF::MF, j < DF::MF);
            }
            load_lds(vin[1], b1 + lk * N + j, DF::MF, j < DF::MF);
            __syncthreads();                                          // B1
            if (!nofft) fft_chain<P, false, 2>(vin, kk, j, ex2, twf); else chain_barriers_only<C::CB>();
            const float rzz = 1.0f / s_rmax[2 * (n & 1)], rxz = 1.0f / s_rmax[2 * (n & 1) + 1];   // (the loader wave filed them with the tile)
                const float sg = ((k + (int)j) & 1) ? -1.f : 1.f;
#pragma unroll
            for (int q = 0; q < DF::RL; ++q) {
                const cf2 den = mk2(kk[0][q].x * rzz + a.lambda, kk[0][q].y * rzz);
                const cf2 num = mk2(kk[1][q].x * rxz, kk[1][q].y * rxz);
                const float inv = ((DF::ML % 2 == 0 || q % 2 == 0) ? sg : -sg) / (den.x * den.x + den.y * den.y);      // IEEE division, as the reference divides (correlation_flow.cc:171)
                const cf2 gg = cmulc(num, den);
                g[0][q] = mk2(gg.x * inv, gg.y * inv);
            }
            if (!(valid0 && j < DF::ML)) zero_fill(g[0]);
            __syncthreads();                                          // B2
            if (!nofft) fft_chain<P, true, 1, WLB>(g, o, j, ex1, twi); else chain_barriers_only<C::CB>();
            if (vst && j < DI::ML) store_strided(o[0], a.dst + (size_t)item * a.dst_stride + loff, DI::ML);
        }
    }
}

// $NIK_RING: bit mask of the B kernels that run in ring form -- 1 fwd_abs_inv, 2 mul_inv / fwd_mul_inv, 4 solve_inv; batches
// too small to give every resident workgroup two tiles keep the one-tile-per-workgroup kernels
#ifndef KCC_RING_DEFAULT
#define KCC_RING_DEFAULT 0
#endif
static int ring_mask() { static const int m = tune_env("NIK_RING") ? atoi(tune_env("NIK_RING")) : KCC_RING_DEFAULT; return m; }
static int ring_wgpc() { static const int m = tune_env("NIK_RING_WGPC") ? atoi(tune_env("NIK_RING_WGPC")) : 0; return m; }
template <int N, int MODE> static bool launchBr_t(hipStream_t s, int n_items, const BArgs& a_in) {
    if constexpr (ring_mode_ok(MODE) && ring_lk(N, MODE) > 0) {
        using C = RCfg<N, MODE>;
        const int bit = MODE == B_FWD_ABS_INV ? 1 : MODE == B_SOLVE_INV ? 4 : 2;
        if (!(ring_mask() & bit)) return false;
        BArgs a = a_in;
        if (C::ALT) { a.tw_f = a.twA_f; a.tw_i = a.twA_i; }
        static const int cus = [] { int dev = 0; hipDeviceProp_t p; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256; return p.multiProcessorCount; }();
        const int wgpc = ring_wgpc() > 0 ? std::min(ring_wgpc(), C::WGPC) : C::WGPC;
        static const bool force = tune_env("NIK_RING_FORCE") && atoi(tune_env("NIK_RING_FORCE"));   // tests: ring form at every batch size
        const int total = ((a.hr + C::LK - 1) / C::LK) * n_items;
        int slots = wgpc * cus;
        if (total < 2 * slots) { if (!force) return false; slots = std::max(1, std::min(slots, (total + 1) / 2)); }
        static const bool big_lds = (C::BYTES > 65536) &&
            (hipFuncSetAttribute(reinterpret_cast<const void*>(&kBr<N, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::BYTES) == hipSuccess);
        (void)big_lds;
        hipLaunchKernelGGL((kBr<N, MODE>), dim3(slots), dim3(C::NT), C::BYTES, s, a, n_items);
        return true;
    } else {
        return false;
    }
}
#else
template <int N, int MODE> static bool launchBr_t(hipStream_t, int, const BArgs&) { return false; }
#endif

template <int N, int MODE> static void launchB_t(hipStream_t s, int n_items, const BArgs& a_in) {
    if (launchBr_t<N, MODE>(s, n_items, a_in)) return;
    BArgs a = a_in;
    if (BCfg<N, MODE>::ALT) { a.tw_f = a.twA_f; a.tw_i = a.twA_i; }
    constexpr int LK = BCfg<N, MODE>::LK;
    dim3 grid((a.hr + LK - 1) / LK, n_items), block(BCfg<N, MODE>::NT);
    constexpr size_t BYTES = BCfg<N, MODE>::BYTES;
    static const bool big_lds = (BYTES > 65536) &&
        (hipFuncSetAttribute(reinterpret_cast<const void*>(&kB<N, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BYTES) == hipSuccess);
    (void)big_lds;
#ifdef KCC_ABLATE
    static const size_t lds_pad = tune_env("NIK_LDS_PAD_B") ? (size_t)atoi(tune_env("NIK_LDS_PAD_B")) : 0;   // occupancy experiments
    hipLaunchKernelGGL((kB<N, MODE>), grid, block, std::min<size_t>(BYTES + lds_pad, 65536), s, a);
    return;
#endif
    hipLaunchKernelGGL((kB<N, MODE>), grid, block, BYTES, s, a);
}

#define DISPATCH_LINE(n, CALL)             \
    switch (n) {                           \
        case 80:   { CALL(80);   break; }  \
        case 160:  { CALL(160);  break; }  \
        case 320:  { CALL(320);  break; }  \
        case 448:  { CALL(448);  break; }  \
        case 480:  { CALL(480);  break; }  \
        case 512:  { CALL(512);  break; }  \
        case 640:  { CALL(640);  break; }  \
        case 752:  { CALL(752);  break; }  \
        case 848:  { CALL(848);  break; }  \
        case 1024: { CALL(1024); break; }  \
        case 1280: { CALL(1280); break; }  \
        case 1600: { CALL(1600); break; }  \
        default: break;                    \
    }

static BArgs base_bargs(PlaneGeom g, Tables t) {
    BArgs a{};
    a.cols = g.cols; a.hr = g.hr; a.tw_f = t.cols_f; a.tw_i = t.cols_i; a.twA_f = t.colsA_f; a.twA_i = t.colsA_i; a.ablate = ablate_flags(); a.rev = g_launch_rev;
    return a;
}

void launch_B_fwd(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* src, size_t src_stride,
                  float2* dst_base, size_t dst_stride, const int* dst_slot) {
    BArgs a = base_bargs(g, t);
    a.src = src; a.src_stride = src_stride; a.dst = dst_base; a.dst_stride = dst_stride; a.dst_slot = dst_slot;
#define CALL(N) launchB_t<N, B_FWD>(s, n_items, a)
    DISPATCH_LINE(g.cols, CALL)
#undef CALL
}
void launch_B_inv(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* src, size_t src_stride,
                  float2* dst, size_t dst_stride) {
    BArgs a = base_bargs(g, t);
    a.src = src; a.src_stride = src_stride; a.dst = dst; a.dst_stride = dst_stride;
#define CALL(N) launchB_t<N, B_INV>(s, n_items, a)
    DISPATCH_LINE(g.cols, CALL)
#undef CALL
}
void launch_B_fwd_abs_inv(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* src, size_t src_stride,
                          float2* dstF_base, size_t dstF_stride, const int* dst_slot, float2* tmp, size_t tmp_stride, int need_cols) {
    BArgs a = base_bargs(g, t);
    a.src = src; a.src_stride = src_stride; a.dst = dstF_base; a.dst_stride = dstF_stride; a.dst_slot = dst_slot;
    a.dst2 = tmp; a.dst2_stride = tmp_stride; a.zz_half = need_cols > 0 ? shifted_columns(g, need_cols) : 0;
#define CALL(N) launchB_t<N, B_FWD_ABS_INV>(s, n_items, a)
    DISPATCH_LINE(g.cols, CALL)
#undef CALL
}
void launch_B_mul_inv(hipStream_t s, int n_items, PlaneGeom g, Tables t, bool x_fwd,
                      const float2* xsrc, size_t x_stride, const int* x_idx,
                      const float2* zsrc, size_t z_stride, const int* z_idx,
                      float2* out, size_t item_stride, size_t plane_stride, unsigned* maxbuf_zero,
                      float2* xstore, size_t xstore_stride, const int* xstore_slot, bool zz_half) {
    BArgs a = base_bargs(g, t);
    a.dst2 = x_fwd ? xstore : nullptr; a.dst2_stride = xstore_stride; a.dst2_slot = xstore_slot;
    a.maxbuf_zero = maxbuf_zero;
    a.zz_half = zz_half ? zz_half_columns(g) : 0;
    a.src = xsrc; a.src_stride = x_stride; a.src_idx = x_idx; a.zsrc = zsrc; a.z_stride = z_stride; a.z_idx = z_idx;
    a.dst = out; a.dst_stride = item_stride; a.out_plane_stride = plane_stride;
    if (x_fwd) {
#define CALL(N) launchB_t<N, B_FWD_MUL_INV>(s, n_items, a)
        DISPATCH_LINE(g.cols, CALL)
#undef CALL
    } else {
#define CALL(N) launchB_t<N, B_MUL_INV>(s, n_items, a)
        DISPATCH_LINE(g.cols, CALL)
#undef CALL
    }
}
void launch_B_zz_inv(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* zsrc, size_t z_stride, const int* z_idx,
                     float2* out, size_t item_stride, unsigned* maxbuf_zero) {
    BArgs a = base_bargs(g, t);
    a.zsrc = zsrc; a.z_stride = z_stride; a.z_idx = z_idx; a.dst = out; a.dst_stride = item_stride; a.maxbuf_zero = maxbuf_zero;
#define CALL(N) launchB_t<N, B_ZZ_INV>(s, n_items, a)
    DISPATCH_LINE(g.cols, CALL)
#undef CALL
}
void launch_B_mul_inv_x(hipStream_t s, int n_items, PlaneGeom g, Tables t, bool x_fwd,
                      const float2* xsrc, size_t x_stride, const int* x_idx,
                      const float2* zsrc, size_t z_stride, const int* z_idx,
                      float2* out, size_t item_stride, size_t plane_stride, unsigned* maxbuf_zero,
                      float2* xstore, size_t xstore_stride, const int* xstore_slot) {
    BArgs a = base_bargs(g, t);
    a.dst2 = x_fwd ? xstore : nullptr; a.dst2_stride = xstore_stride; a.dst2_slot = xstore_slot;
    a.src = xsrc; a.src_stride = x_stride; a.src_idx = x_idx; a.zsrc = zsrc; a.z_stride = z_stride; a.z_idx = z_idx;
    a.dst = out; a.dst_stride = item_stride; a.out_plane_stride = plane_stride; a.maxbuf_zero = maxbuf_zero;
    if (x_fwd) {
#define CALL(N) launchB_t<N, B_FWD_MUL_INV_X>(s, n_items, a)
        DISPATCH_LINE(g.cols, CALL)
#undef CALL
    } else {
#define CALL(N) launchB_t<N, B_MUL_INV_X>(s, n_items, a)
        DISPATCH_LINE(g.cols, CALL)
#undef CALL
    }
}
void launch_B_solve_cached(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* buf, size_t item_stride,
                           size_t plane_stride, const unsigned* maxbuf, const float2* kzz, size_t kzz_stride,
                           const unsigned* mzz, const int* z_idx, float lambda, float2* out, size_t out_stride) {
    BArgs a = base_bargs(g, t);
    a.src = buf; a.src_stride = item_stride; a.in_plane_stride = plane_stride; a.maxbuf = maxbuf; a.lambda = lambda;
    a.n_parts[0] = 0; a.n_parts[1] = kfwd_parts(g, false);
    a.kzz = kzz; a.kzz_stride = kzz_stride; a.mzz = mzz; a.z_idx = z_idx; a.dst = out; a.dst_stride = out_stride;
#define CALL(N) launchB_t<N, B_SOLVE_CACHED>(s, n_items, a)
    DISPATCH_LINE(g.cols, CALL)
#undef CALL
}
__global__ __launch_bounds__(64) void k_store_mzz(const unsigned* __restrict__ maxbuf, int n_parts, const int* __restrict__ slots, unsigned* __restrict__ mzz) {
    const int i = blockIdx.x;
    const float m = parts_max(maxbuf + (size_t)(2 * i) * KCC_MAXPARTS, n_parts, threadIdx.x);
    if (threadIdx.x == 0) mzz[slots[i]] = __float_as_uint(m);
}
void launch_store_mzz(hipStream_t s, int n, PlaneGeom g, const unsigned* maxbuf, const int* slots, unsigned* mzz) {
    hipLaunchKernelGGL(k_store_mzz, dim3(n), dim3(64), 0, s, maxbuf, kfwd_parts(g, false), slots, mzz);
}
void launch_B_solve_inv(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* buf, size_t item_stride,
                        size_t plane_stride, const unsigned* maxbuf, float lambda, float2* out, size_t out_stride, bool zz_half) {
    BArgs a = base_bargs(g, t);
    a.src = buf; a.src_stride = item_stride; a.in_plane_stride = plane_stride; a.maxbuf = maxbuf; a.lambda = lambda;
    a.n_parts[0] = kfwd_parts(g, zz_half); a.n_parts[1] = kfwd_parts(g, false);
    a.dst = out; a.dst_stride = out_stride; a.zz_half = zz_half ? 1 : 0;
#define CALL(N) launchB_t<N, B_SOLVE_INV>(s, n_items, a)
    DISPATCH_LINE(g.cols, CALL)
#undef CALL
}

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
__global__ void k_energy(CCP xsrc, size_t x_stride, const int* x_idx,
                         CCP zsrc, size_t z_stride, const int* z_idx, size_t n, float* energy) {
    __shared__ double red[256];
    const int item = blockIdx.x, which = blockIdx.y, tid = threadIdx.x;
    const cf2* p = which == 0 ? xsrc + (size_t)(x_idx ? x_idx[item] : item) * x_stride
                                 : zsrc + (size_t)(z_idx ? z_idx[item] : item) * z_stride;
    double acc = 0.0;
    for (size_t i = tid; i < n; i += 256) { const cf2 v = p[i]; acc += (double)(v.x * v.x + v.y * v.y); }
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    if (tid == 0) energy[2 * item + which] = (float)red[0];
}
void launch_energy(hipStream_t s, int n_items, PlaneGeom g, const float2* xsrc, size_t x_stride, const int* x_idx,
                   const float2* zsrc, size_t z_stride, const int* z_idx, float* energy) {
    hipLaunchKernelGGL(k_energy, dim3(n_items, 2), dim3(256), 0, s, CCP(xsrc), x_stride, x_idx, CCP(zsrc), z_stride, z_idx,
                       (size_t)g.hr * g.cols, energy);
}

// reduce the per-workgroup partials of one response surface (fixed order -> deterministic); for the rotation
// surface also emit the de-rotation table index of every translation item of the pair:
//   rot_index[item*n_hyp + h] = variant(h) * PD + arg-max row       (variant 0: small-rotation fold, 1/2: orig / +180)
__global__ __launch_bounds__(64) void k_finalize(const Partial* __restrict__ partials, int partial_stride, int n_partials,
                                                 SurfaceResult* out, int* rot_index, int n_hyp, int PD, SurfaceResult* host_out) {
    const int item = blockIdx.x, lane = threadIdx.x;
    const Partial* p = partials + (size_t)item * partial_stride;
    // one wave: lane i folds partials i, i+64, ... in index order, then a fixed butterfly -> deterministic
    double s1 = 0, s2 = 0; float peak = -INFINITY; int idx = 0x7FFFFFFF;
    for (int i = lane; i < n_partials; i += 64) {
        s1 += p[i].sum; s2 += p[i].sumsq;
        if (p[i].peak > peak || (p[i].peak == peak && p[i].idx < idx)) { peak = p[i].peak; idx = p[i].idx; }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const float op = __shfl_xor(peak, off); const int oi = __shfl_xor(idx, off);
        if (op > peak || (op == peak && oi < idx)) { peak = op; idx = oi; }
        s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off);
    }
    if (lane == 0) {
        SurfaceResult r; r.sum = s1; r.sumsq = s2; r.peak = peak; r.idx = idx;
        out[item] = r;
        if (host_out) host_out[item] = r;                     // pinned host mirror: the result needs no copy of its own
        if (rot_index)
            for (int h = 0; h < n_hyp; ++h) rot_index[item * n_hyp + h] = (n_hyp == 1 ? 0 : 1 + h) * PD + (idx % PD);
    }
}
void launch_finalize(hipStream_t s, int n_items, const Partial* partials, int partial_stride, int n_partials, SurfaceResult* out,
                     int* rot_index, int n_hyp, int PD, SurfaceResult* host_out) {
    hipLaunchKernelGGL(k_finalize, dim3(n_items), dim3(64), 0, s, partials, partial_stride, n_partials, out, rot_index, n_hyp, PD, host_out);
}

// Residual statistics of one batch call, on the device (north star: "RCCL all-reduce only for the final pose-graph residual
// sum"): stats = [sum PSR_t (chosen hypothesis), sum PSR_r, sum |t|^2 (px^2), count] from the raw surface results, with
// GetInfo (correlation_flow.cc:238-243) and the hypothesis choice (:121-131) evaluated as finalize_pose does on the host.
// One wave, fixed summation order: deterministic.
__device__ __forceinline__ float psr_dev(const SurfaceResult& r, long n) {
    const double m = (r.sum - (double)r.peak) / (double)(n - 1);
    double var = (r.sumsq - 2.0 * m * r.sum + (double)n * m * m) / (double)n;
    if (var < 0) var = 0;
    return (float)(((double)r.peak - m) / ((double)(float)sqrt(var) + 1e-7));
}
__global__ __launch_bounds__(64) void k_residual_stats(const SurfaceResult* __restrict__ rot, const SurfaceResult* __restrict__ trans,
                                                       int n, int n_hyp, int H, int W, int PD, int PC, double* __restrict__ stats) {
    double s[4] = { 0, 0, 0, 0 };
    for (int i = threadIdx.x; i < n; i += 64) {
        int h = 0;
        float pt = psr_dev(trans[i * n_hyp], (long)H * W);
        if (n_hyp == 2) { const float p1 = psr_dev(trans[i * 2 + 1], (long)H * W); if (!(pt > p1)) { pt = p1; h = 1; } }
        const int idx = trans[i * n_hyp + h].idx;
        const double t0 = -((idx % H) - H / 2), t1 = -((idx / H) - W / 2);
        s[0] += (double)pt; s[1] += (double)psr_dev(rot[i], (long)PD * PC); s[2] += t0 * t0 + t1 * t1; s[3] += 1.0;
    }
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += __shfl_xor(s[k], off);
    if (threadIdx.x == 0) { stats[0] = s[0]; stats[1] = s[1]; stats[2] = s[2]; stats[3] = s[3]; }
}
void launch_residual_stats(hipStream_t st, const SurfaceResult* rot, const SurfaceResult* trans, int n, int n_hyp,
                           int H, int W, int PD, int PC, double* stats) {
    hipLaunchKernelGGL(k_residual_stats, dim3(1), dim3(64), 0, st, rot, trans, n, n_hyp, H, W, PD, PC, stats);
}
// total[0..3] = sum over `n_parts` partial blocks of 4 doubles, in index order
__global__ void k_stats_sum(const double* __restrict__ parts, int n_parts, double* __restrict__ total) {
    const int k = threadIdx.x;
    if (k < 4) { double a = 0; for (int p = 0; p < n_parts; ++p) a += parts[4 * p + k]; total[k] = a; }
}
void launch_stats_sum(hipStream_t st, const double* parts, int n_parts, double* total) {
    hipLaunchKernelGGL(k_stats_sum, dim3(1), dim3(4), 0, st, parts, n_parts, total);
}

// Coarse-to-fine chaining on the device (BASELINE config 3, an extension): the arg-max window centres of this level predicted
// from the surface results of the level above -- offsets from the surface centre scale with the surface size, rounded half
// away from zero (kcc_pyramid.cpp predict()), wrapped cyclically.
__device__ __forceinline__ int predict_idx(int idx, int n_from, int n_to) {
    const double off = (double)(idx - n_from / 2) * (double)n_to / (double)n_from;
    long p = n_to / 2 + lround(off);
    p %= n_to; if (p < 0) p += n_to;
    return (int)p;
}
__global__ void k_predict_windows(const SurfaceResult* __restrict__ rot, const SurfaceResult* __restrict__ trans, int n,
                                  int PDu, int PCu, int Hu, int Wu, int PD, int PC, int H, int W,
                                  int* __restrict__ wrr, int* __restrict__ wrc, int* __restrict__ wtr, int* __restrict__ wtc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ri = rot[i].idx, ti = trans[i].idx;
    wrr[i] = predict_idx(ri % PDu, PDu, PD); wrc[i] = predict_idx(ri / PDu, PCu, PC);
    wtr[i] = predict_idx(ti % Hu, Hu, H);    wtc[i] = predict_idx(ti / Hu, Wu, W);
}
void launch_predict_windows(hipStream_t s, const SurfaceResult* rot, const SurfaceResult* trans, int n, int PDu, int PCu, int Hu, int Wu,
                            int PD, int PC, int H, int W, int* wrr, int* wrc, int* wtr, int* wtc) {
    hipLaunchKernelGGL(k_predict_windows, dim3((n + 63) / 64), dim3(64), 0, s, rot, trans, n, PDu, PCu, Hu, Wu, PD, PC, H, W, wrr, wrc, wtr, wtc);
}

// RemoveZeroComponent (correlation_flow.cc:79-87) on the shifted plane S: p(r,c) lives at S[(c+W/2)%W][(r+H/2)%H].
//   column c=0 (all r):  (p(r,1) + p(r,W-1))/2   -- reads the ORIGINAL columns 1 and W-1, so it runs first
//   row r=0 (c != 0):    (p(1,c) + p(H-1,c))/2
__global__ void k_fix_zero(float* __restrict__ Sbase, size_t s_stride, int H, int W) {
    float* S = Sbase + (size_t)blockIdx.x * s_stride;
    const int SP = H + 2, hx = W / 2, hy = H / 2;
    const int x1 = (1 + hx) % W, xw = (W - 1 + hx) % W;
    for (int ys = threadIdx.x; ys < H; ys += blockDim.x)
        S[(size_t)hx * SP + ys] = (S[(size_t)x1 * SP + ys] + S[(size_t)xw * SP + ys]) * 0.5f;
    __syncthreads();
    const int y1 = (1 + hy) % H, yh = (H - 1 + hy) % H;
    for (int xs = threadIdx.x; xs < W; xs += blockDim.x)
        if (xs != hx) S[(size_t)xs * SP + hy] = (S[(size_t)xs * SP + y1] + S[(size_t)xs * SP + yh]) * 0.5f;
}
void launch_fix_zero(hipStream_t s, int n_items, float* S, size_t s_stride, int H, int W) {
    hipLaunchKernelGGL(k_fix_zero, dim3(n_items), dim3(256), 0, s, S, s_stride, H, W);
}
// debug: plain plane p (column-major H x W) -> shifted plane S (no zero-component fix)
__global__ void k_make_shifted(const float* __restrict__ p, float* __restrict__ S, int H, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < H * W) {
        const int r = i % H, c = i / H;
        S[(size_t)((c + W / 2) % W) * (H + 2) + (r + H / 2) % H] = p[i];
    }
}
void launch_make_shifted(hipStream_t s, const float* p, float* S, int H, int W) {
    hipLaunchKernelGGL(k_make_shifted, dim3((H * W + 255) / 256), dim3(256), 0, s, p, S, H, W);
}

__global__ void k_transpose_c(CCP src, CP dst, int R, int C) {
    __shared__ cf2 tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        if (r < R && c < C) tile[j][threadIdx.x] = src[(size_t)r * C + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (r < R && c < C) dst[(size_t)c * R + r] = tile[threadIdx.x][j];
    }
}
void launch_transpose_c(hipStream_t s, const float2* src, float2* dst, int src_rows, int src_cols) {
    dim3 grid((src_cols + 31) / 32, (src_rows + 31) / 32), block(32, 8);
    hipLaunchKernelGGL(k_transpose_c, grid, block, 0, s, CCP(src), CP(dst), src_rows, src_cols);
}

}  // namespace kcc
