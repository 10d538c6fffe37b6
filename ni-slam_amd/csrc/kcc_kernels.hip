// kcc_kernels.hip -- gfx950 kernels of the KCC front end.
//
// Two kernel families implement every 2-D real FFT of the path (reference correlation_flow.cc:53-77):
//   A-type: LX lines along the halved axis (rows) per workgroup; real<->half-complex via one complex
//           FFT of length rows/2; the spectrum side is accessed transposed in 128-byte segments.
//   B-type: LK contiguous spectrum lines (length cols) per workgroup.
// Spectra are stored k-major ([rows/2+1][cols], cols contiguous) so the B pass is fully coalesced.
// All pointwise work of the path (|F|, X conj Z, kernel function, ridge solve, max, arg-max, PSR
// moments, polar / rotation gathers) is fused into the load or store side of these passes.
#include "kcc_kernels.h"
#include "kcc_fft.h"

namespace kcc {

// ------------------------------------------------------------------------------------------------
// instantiated FFT lengths
// ------------------------------------------------------------------------------------------------
#define KCC_HALF_LIST(X) X(30) X(60) X(120) X(240) X(360)
#define KCC_LINE_LIST(X) X(80) X(160) X(320) X(480) X(640) X(1280)

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

constexpr int A_LX = 16;     // lines per A-type workgroup (16 float2 = one 128-B segment per spectrum row)
constexpr int A_NT = 256;
constexpr int B_NT = 256;

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wrap_idx(int p, int len) {      // cv::borderInterpolate(BORDER_WRAP)
    if (p < 0) p -= ((p - len + 1) / len) * len;
    if (p >= len) p %= len;
    return p;
}

// cv::remap bilinear weights from 1/32-pixel fractions, summed in OpenCV's order.  Contraction is
// switched off so the arithmetic is the same mul/add sequence the CPU executes (bit-exact gathers).
__device__ __forceinline__ float bilerp(float v0, float v1, float v2, float v3, int fx, int fy) {
#pragma clang fp contract(off)
    const float s = 1.f / 32.f;
    const float tx1 = (float)fx * s, tx0 = 1.f - tx1;
    const float ty1 = (float)fy * s, ty0 = 1.f - ty1;
    const float w0 = ty0 * tx0, w1 = ty0 * tx1, w2 = ty1 * tx0, w3 = ty1 * tx1;
    float acc = v0 * w0;
    acc = acc + v1 * w1;
    acc = acc + v2 * w2;
    acc = acc + v3 * w3;
    return acc;
}
// saturate_cast<int>(m*c*AB_SCALE) and saturate_cast<int>((m1*r + m2)*AB_SCALE) of cv::warpAffine, in double
__device__ __forceinline__ int affine_delta(double m, int c) {
#pragma clang fp contract(off)
    const double t = m * (double)c;
    return __double2int_rn(t * 1024.0);
}
__device__ __forceinline__ int affine_base(double m1, int r, double m2) {
#pragma clang fp contract(off)
    const double t = m1 * (double)r;
    const double u = t + m2;
    return __double2int_rn(u * 1024.0);
}

// (xz + offset)^power as Eigen's Array::pow(int) does it: double pow, rounded to float.
__device__ __forceinline__ float kernel_value(const KernelFn& fn, float xz, float gauss_bias, float gauss_scale) {
    if (fn.type == 0) {
        const double b = (double)(xz + fn.offset);
        double r;
        if (fn.power == 3) r = b * b * b;
        else if (fn.power == 2) r = b * b;
        else if (fn.power == 1) r = b;
        else r = pow(b, (double)fn.power);
        return (float)r;
    }
    // gaussian: exp(-1/sigma^2 * (xx + zz - 2 xz)/N)   (correlation_flow.cc:189-190)
    const float xxzz = (gauss_bias - 2.f * xz) * gauss_scale;       // gauss_scale = 1/N (as division below)
    return __expf(xxzz);
}

// ------------------------------------------------------------------------------------------------
// u8 row-major -> f32 column-major, /255  (utils.cc:110-118)
// ------------------------------------------------------------------------------------------------
__global__ void k_cvt_u8(const uint8_t* __restrict__ src, const int* __restrict__ dst_slot, float* __restrict__ arena,
                         int H, int W) {
    __shared__ float tile[32][33];
    const int item = blockIdx.z;
    const uint8_t* in = src + (size_t)item * H * W;
    float* out = arena + (size_t)dst_slot[item] * H * W;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        if (r < H && c < W) tile[j][threadIdx.x] = (float)in[(size_t)r * W + c] / 255.0f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (r < H && c < W) out[(size_t)c * H + r] = tile[threadIdx.x][j];
    }
}

void launch_cvt_u8(hipStream_t s, int n, const uint8_t* d_gray, const int* d_dst_slot, float* arena_img, int H, int W) {
    dim3 grid((W + 31) / 32, (H + 31) / 32, n), block(32, 8);
    hipLaunchKernelGGL(k_cvt_u8, grid, block, 0, s, d_gray, d_dst_slot, arena_img, H, W);
}

// ------------------------------------------------------------------------------------------------
// A-type kernels
// ------------------------------------------------------------------------------------------------
enum { SRC_PLANE = 0, SRC_ROT = 1, SRC_POLAR = 2 };
enum { EPI_REAL = 0, EPI_KERNEL_FWD = 1, EPI_ARGMAX = 2 };

struct AArgs {
    // geometry of the transformed plane
    int rows, cols, hr;
    const float2* tw_half;
    const float2* tw_full;
    // forward source
    const float* src; size_t src_stride; const int* src_idx;
    // rotation source
    const RotEntry* rot_tab; const int* rot_index;
    // polar source (src = p planes, H x W)
    int H, W; const uint32_t* polar_tab;
    // spectrum side
    float2* spec; size_t spec_stride; size_t plane_stride;
    // inverse outputs
    float* real_out; size_t real_stride;
    Partial* partials; int partial_stride;
    KernelFn fn; unsigned* maxbuf; const float* energy;
};

// LDS layout of one A workgroup: LX lines x (h+1) float2 (pitch odd -> conflict-free transposes), + W_h table
template <int HH> struct ALds {
    static constexpr int PITCH = HH + 1;
    static constexpr int DATA = A_LX * PITCH;
    static constexpr size_t BYTES = (size_t)(DATA + HH) * sizeof(float2);
};

template <int HH>
__device__ __forceinline__ void a_load_tw(float2* tw_lds, const float2* __restrict__ tw_half, int tid) {
    for (int i = tid; i < HH; i += A_NT) tw_lds[i] = tw_half[i];
}

// spectrum tile [k][x0..x0+LX) (global, k-major)  ->  lds[xx][k]
template <int HH>
__device__ __forceinline__ void a_load_spec(float2* lds, const float2* __restrict__ spec, int cols, int x0, int tid) {
    constexpr int PITCH = ALds<HH>::PITCH;
    for (int idx = tid; idx < (HH + 1) * A_LX; idx += A_NT) {
        const int xx = idx % A_LX, k = idx / A_LX;
        lds[xx * PITCH + k] = spec[(size_t)k * cols + x0 + xx];
    }
}
template <int HH>
__device__ __forceinline__ void a_store_spec(const float2* lds, float2* __restrict__ spec, int cols, int x0, int tid) {
    constexpr int PITCH = ALds<HH>::PITCH;
    for (int idx = tid; idx < (HH + 1) * A_LX; idx += A_NT) {
        const int xx = idx % A_LX, k = idx / A_LX;
        spec[(size_t)k * cols + x0 + xx] = lds[xx * PITCH + k];
    }
}

// Z (FFT of the packed line) -> X[0..h] in place  (r2c split)
template <int HH>
__device__ __forceinline__ void a_r2c_post(float2* lds, const float2* __restrict__ tw_full, int tid) {
    constexpr int PITCH = ALds<HH>::PITCH;
    constexpr int NP = HH / 2 + 1;
    for (int idx = tid; idx < A_LX * NP; idx += A_NT) {
        const int line = idx / NP, k = idx - line * NP;
        float2* L = lds + line * PITCH;
        if (k == 0) {
            const float2 z = L[0];
            L[0] = make_float2(z.x + z.y, 0.f);
            L[HH] = make_float2(z.x - z.y, 0.f);
        } else {
            const float2 a = L[k], b = cconj(L[HH - k]);
            const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
            const float2 d = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
            const float2 o = make_float2(d.y, -d.x);            // -i d
            const float2 t = cmul(tw_full[k], o);
            L[k] = cadd(e, t);
            if (2 * k != HH) L[HH - k] = cconj(csub(e, t));
        }
    }
}
// X[0..h] -> Z' (input of the inverse packed FFT), in place  (c2r merge; imag of X[0], X[h] ignored like FFTW)
template <int HH>
__device__ __forceinline__ void a_c2r_pre(float2* lds, const float2* __restrict__ tw_full, int tid) {
    constexpr int PITCH = ALds<HH>::PITCH;
    constexpr int NP = HH / 2 + 1;
    for (int idx = tid; idx < A_LX * NP; idx += A_NT) {
        const int line = idx / NP, k = idx - line * NP;
        float2* L = lds + line * PITCH;
        if (k == 0) {
            const float x0 = L[0].x, xh = L[HH].x;
            L[0] = make_float2(x0 + xh, x0 - xh);
        } else {
            const float2 a = L[k], b = cconj(L[HH - k]);
            const float2 sm = cadd(a, b), d = csub(a, b);
            const float2 u = cmulc(d, tw_full[k]);               // conj(w^k) * d
            const float2 iu = make_float2(-u.y, u.x);
            L[k] = cadd(sm, iu);
            if (2 * k != HH) L[HH - k] = cconj(csub(sm, iu));
        }
    }
}

// value of fftshift(RemoveZeroComponent(p)) at shifted coordinates (y, x), zero outside
// (correlation_flow.cc:79-87,93-94; circ_shift.h:131-154,238-244)
__device__ __forceinline__ float shifted_hp(const float* __restrict__ p, int H, int W, int y, int x) {
    if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) return 0.f;
    int r = y - H / 2; if (r < 0) r += H;
    int c = x - W / 2; if (c < 0) c += W;
    if (c == 0) return (p[(size_t)1 * H + r] + p[(size_t)(W - 1) * H + r]) * 0.5f;
    if (r == 0) return (p[(size_t)c * H + 1] + p[(size_t)c * H + H - 1]) * 0.5f;
    return p[(size_t)c * H + r];
}

// RotateArray (utils.cc:154-161): cv::warpAffine(INTER_LINEAR, BORDER_WRAP) with the inverse matrix of the
// candidate angle; fixed-point coordinates exactly as OpenCV's WarpAffineInvoker.  dst pixel (r, c).
__device__ __forceinline__ float rot_sample(const float* __restrict__ img, int H, int W, const RotEntry& R, int r, int c) {
    const int adelta = affine_delta(R.m[0], c);
    const int bdelta = affine_delta(R.m[3], c);
    const int X0 = affine_base(R.m[1], r, R.m[2]) + 16;
    const int Y0 = affine_base(R.m[4], r, R.m[5]) + 16;
    const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    int sx = X >> 5, sy = Y >> 5;
    sx = max(-32768, min(32767, sx)); sy = max(-32768, min(32767, sy));
    const int xa = wrap_idx(sx, W), xb = wrap_idx(sx + 1, W);
    const int ya = wrap_idx(sy, H), yb = wrap_idx(sy + 1, H);
    return bilerp(img[(size_t)xa * H + ya], img[(size_t)xb * H + ya],
                  img[(size_t)xa * H + yb], img[(size_t)xb * H + yb], X & 31, Y & 31);
}
// polar(fftshift(RemoveZeroComponent(p))) (correlation_flow.cc:228-236): one table-driven remap sample
__device__ __forceinline__ float polar_sample(const float* __restrict__ p, int H, int W, uint32_t t) {
    const int sx = t & 0x7FF, sy = (t >> 11) & 0x7FF, fx = (t >> 22) & 31, fy = t >> 27;
    return bilerp(shifted_hp(p, H, W, sy, sx), shifted_hp(p, H, W, sy, sx + 1),
                  shifted_hp(p, H, W, sy + 1, sx), shifted_hp(p, H, W, sy + 1, sx + 1), fx, fy);
}

template <int HH, int SRC>
__global__ __launch_bounds__(A_NT) void kA_fwd(AArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PITCH = ALds<HH>::PITCH;
    float2* lds = reinterpret_cast<float2*>(smem);
    float2* tw = lds + ALds<HH>::DATA;
    const int tid = threadIdx.x, item = blockIdx.y, x0 = blockIdx.x * A_LX;
    a_load_tw<HH>(tw, a.tw_half, tid);

    if (SRC == SRC_PLANE) {
        const int pl = a.src_idx ? a.src_idx[item] : item;
        const float2* src = reinterpret_cast<const float2*>(a.src + (size_t)pl * a.src_stride + (size_t)x0 * a.rows);
        for (int idx = tid; idx < A_LX * HH; idx += A_NT) {
            const int line = idx / HH, m = idx - line * HH;
            lds[line * PITCH + m] = src[(size_t)line * HH + m];
        }
    } else if (SRC == SRC_ROT) {
        const int H = a.rows, W = a.cols;
        const float* img = a.src + (size_t)a.src_idx[item] * a.src_stride;
        const RotEntry R = a.rot_tab[a.rot_index[item]];
        for (int idx = tid; idx < A_LX * HH; idx += A_NT) {
            const int line = idx / HH, m = idx - line * HH;
            const int c = x0 + line;
            lds[line * PITCH + m] = make_float2(rot_sample(img, H, W, R, 2 * m, c), rot_sample(img, H, W, R, 2 * m + 1, c));
        }
    } else {
        const int PD = a.rows;
        const float* p = a.src + (size_t)item * a.src_stride;
        for (int idx = tid; idx < A_LX * HH; idx += A_NT) {
            const int line = idx / HH, m = idx - line * HH;
            const uint32_t* tab = a.polar_tab + (size_t)(x0 + line) * PD + 2 * m;
            lds[line * PITCH + m] = make_float2(polar_sample(p, a.H, a.W, tab[0]), polar_sample(p, a.H, a.W, tab[1]));
        }
    }
    __syncthreads();
    line_fft<HH, A_LX, A_NT, PITCH, false>(lds, tw, tid);
    a_r2c_post<HH>(lds, a.tw_full, tid);
    __syncthreads();
    a_store_spec<HH>(lds, a.spec + (size_t)item * a.spec_stride, a.cols, x0, tid);
}

template <int HH, int EPI>
__global__ __launch_bounds__(A_NT) void kA_inv(AArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PITCH = ALds<HH>::PITCH;
    float2* lds = reinterpret_cast<float2*>(smem);
    float2* tw = lds + ALds<HH>::DATA;
    __shared__ float red_f[A_NT / 64];
    __shared__ int red_i[A_NT / 64];
    __shared__ double red_d[2][A_NT / 64];
    const int tid = threadIdx.x, item = blockIdx.y, x0 = blockIdx.x * A_LX;
    const int plane = (EPI == EPI_KERNEL_FWD) ? blockIdx.z : 0;
    float2* spec = a.spec + (size_t)item * a.spec_stride + (size_t)plane * a.plane_stride;
    a_load_tw<HH>(tw, a.tw_half, tid);
    a_load_spec<HH>(lds, spec, a.cols, x0, tid);
    __syncthreads();
    a_c2r_pre<HH>(lds, a.tw_full, tid);
    __syncthreads();
    line_fft<HH, A_LX, A_NT, PITCH, true>(lds, tw, tid);
    const float size = (float)((long)a.rows * a.cols);           // IFFT: x / x.size()  (correlation_flow.cc:76)

    if (EPI == EPI_REAL) {
        float2* dst = reinterpret_cast<float2*>(a.real_out + (size_t)item * a.real_stride + (size_t)x0 * a.rows);
        for (int idx = tid; idx < A_LX * HH; idx += A_NT) {
            const int line = idx / HH, m = idx - line * HH;
            const float2 z = lds[line * PITCH + m];
            dst[(size_t)line * HH + m] = make_float2((z.x / size), (z.y / size));
        }
    } else if (EPI == EPI_KERNEL_FWD) {
        float gbias = 0.f, gscale = 0.f;
        if (a.fn.type == 1) {
            const float N = size;
            const float xx = a.energy[2 * item + 0] / N, zz = a.energy[2 * item + 1] / N;
            gbias = (plane == 0) ? (zz + zz) : (xx + zz);
            gscale = (-1.f / (a.fn.sigma * a.fn.sigma)) / N;
        }
        float mx = 0.f;
        for (int idx = tid; idx < A_LX * HH; idx += A_NT) {
            const int line = idx / HH, m = idx - line * HH;
            const float2 z = lds[line * PITCH + m];
            const float k0 = kernel_value(a.fn, (z.x / size), gbias, gscale);
            const float k1 = kernel_value(a.fn, (z.y / size), gbias, gscale);
            mx = fmaxf(mx, fmaxf(fabsf(k0), fabsf(k1)));
            lds[line * PITCH + m] = make_float2(k0, k1);
        }
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if ((tid & 63) == 0) red_f[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) {
            float m2 = red_f[0];
            for (int w = 1; w < A_NT / 64; ++w) m2 = fmaxf(m2, red_f[w]);
            atomicMax(a.maxbuf + 2 * item + plane, __float_as_uint(m2));   // non-negative floats order as uints
        }
        line_fft<HH, A_LX, A_NT, PITCH, false>(lds, tw, tid);
        a_r2c_post<HH>(lds, a.tw_full, tid);
        __syncthreads();
        a_store_spec<HH>(lds, spec, a.cols, x0, tid);
    } else {
        // arg-max (column-major first strict max, Eigen maxCoeff visitor) + moments for GetInfo
        float best = -INFINITY; int bidx = 0x7FFFFFFF;
        float s1 = 0.f, s2 = 0.f;
        for (int idx = tid; idx < A_LX * HH; idx += A_NT) {
            const int line = idx / HH, m = idx - line * HH;
            const float2 z = lds[line * PITCH + m];
            const float g0 = (z.x / size), g1 = (z.y / size);
            const int li = (x0 + line) * a.rows + 2 * m;
            if (g0 > best) { best = g0; bidx = li; }
            if (g1 > best) { best = g1; bidx = li + 1; }
            s1 += g0 + g1; s2 += g0 * g0 + g1 * g1;
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
            for (int w = 1; w < A_NT / 64; ++w) {
                if (red_f[w] > best || (red_f[w] == best && red_i[w] < bidx)) { best = red_f[w]; bidx = red_i[w]; }
                d1 += red_d[0][w]; d2 += red_d[1][w];
            }
            Partial p; p.sum = d1; p.sumsq = d2; p.peak = best; p.idx = bidx;
            a.partials[(size_t)item * a.partial_stride + blockIdx.x] = p;
        }
    }
}

int argmax_blocks(PlaneGeom g) { return g.cols / A_LX; }

template <int HH, int SRC> static void launchA_fwd_t(hipStream_t s, int n_items, const AArgs& a) {
    dim3 grid(a.cols / A_LX, n_items), block(A_NT);
    hipLaunchKernelGGL((kA_fwd<HH, SRC>), grid, block, ALds<HH>::BYTES, s, a);
}
template <int HH, int EPI> static void launchA_inv_t(hipStream_t s, int n_items, int nz, const AArgs& a) {
    dim3 grid(a.cols / A_LX, n_items, nz), block(A_NT);
    hipLaunchKernelGGL((kA_inv<HH, EPI>), grid, block, ALds<HH>::BYTES, s, a);
}

static AArgs base_args(PlaneGeom g, Tables t) {
    AArgs a{};
    a.rows = g.rows; a.cols = g.cols; a.hr = g.hr; a.tw_half = t.tw_half; a.tw_full = t.tw_full;
    return a;
}

#define DISPATCH_HALF(h, CALL)            \
    switch (h) {                          \
        case 30:  { CALL(30);  break; }   \
        case 60:  { CALL(60);  break; }   \
        case 120: { CALL(120); break; }   \
        case 240: { CALL(240); break; }   \
        case 360: { CALL(360); break; }   \
        default: break;                   \
    }

void launch_A_fwd_plane(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float* src, size_t src_stride,
                        const int* src_idx, float2* dst, size_t dst_stride) {
    AArgs a = base_args(g, t);
    a.src = src; a.src_stride = src_stride; a.src_idx = src_idx; a.spec = dst; a.spec_stride = dst_stride;
#define CALL(HH) launchA_fwd_t<HH, SRC_PLANE>(s, n_items, a)
    DISPATCH_HALF(g.rows / 2, CALL)
#undef CALL
}
void launch_A_fwd_rot(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float* arena_img, size_t img_stride,
                      const int* img_slot, const RotEntry* rot_tab, const int* rot_index, float2* dst, size_t dst_stride) {
    AArgs a = base_args(g, t);
    a.src = arena_img; a.src_stride = img_stride; a.src_idx = img_slot; a.rot_tab = rot_tab; a.rot_index = rot_index;
    a.spec = dst; a.spec_stride = dst_stride;
#define CALL(HH) launchA_fwd_t<HH, SRC_ROT>(s, n_items, a)
    DISPATCH_HALF(g.rows / 2, CALL)
#undef CALL
}
void launch_A_fwd_polar(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float* p, size_t p_stride,
                        int H, int W, const uint32_t* polar_tab, float2* dst, size_t dst_stride) {
    AArgs a = base_args(g, t);
    a.src = p; a.src_stride = p_stride; a.H = H; a.W = W; a.polar_tab = polar_tab; a.spec = dst; a.spec_stride = dst_stride;
#define CALL(HH) launchA_fwd_t<HH, SRC_POLAR>(s, n_items, a)
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
void launch_A_inv_kernel_fwd(hipStream_t s, int n_items, PlaneGeom g, Tables t, float2* buf, size_t item_stride,
                             size_t plane_stride, KernelFn fn, unsigned* maxbuf, const float* energy) {
    AArgs a = base_args(g, t);
    a.spec = buf; a.spec_stride = item_stride; a.plane_stride = plane_stride; a.fn = fn; a.maxbuf = maxbuf; a.energy = energy;
#define CALL(HH) launchA_inv_t<HH, EPI_KERNEL_FWD>(s, n_items, 2, a)
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
enum { B_FWD = 0, B_FWD_ABS_INV = 1, B_MUL_INV = 2, B_FWD_MUL_INV = 3, B_SOLVE_INV = 4, B_INV = 5 };

struct BArgs {
    int cols, hr;
    const float2* tw_cols;
    const float2* src; size_t src_stride; const int* src_idx;     // primary input
    const float2* zsrc; size_t z_stride; const int* z_idx;        // Z (key) spectra
    size_t in_plane_stride;                                       // SOLVE: plane 1 offset inside src item
    float2* dst; size_t dst_stride; const int* dst_slot;          // primary output
    float2* dst2; size_t dst2_stride;                             // secondary output
    size_t out_plane_stride;                                      // MUL_INV: plane 1 offset inside dst item
    const unsigned* maxbuf; float lambda;
};

template <int N> struct BCfg {
    static constexpr int LK = (N <= 160) ? 8 : (N <= 640 ? 2 : 1);     // lines per set
    static constexpr size_t BYTES = (size_t)(2 * LK * N + N) * sizeof(float2);
};

template <int N, int MODE>
__global__ __launch_bounds__(B_NT) void kB(BArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LK = BCfg<N>::LK;
    float2* set0 = reinterpret_cast<float2*>(smem);
    float2* set1 = set0 + LK * N;
    float2* tw = set1 + LK * N;
    const int tid = threadIdx.x, item = blockIdx.y, k0 = blockIdx.x * LK;
    const int nl = min(LK, a.hr - k0);                               // valid lines in this workgroup
    for (int i = tid; i < N; i += B_NT) tw[i] = a.tw_cols[i];

    const size_t line0 = (size_t)k0 * N;
    // ---- stage 1: primary load
    if (MODE == B_FWD || MODE == B_FWD_ABS_INV || MODE == B_FWD_MUL_INV) {
        const int pl = a.src_idx ? a.src_idx[item] : item;
        const float2* src = a.src + (size_t)pl * a.src_stride + line0;
        float2* dstset = (MODE == B_FWD_MUL_INV) ? set1 : set0;
        for (int i = tid; i < LK * N; i += B_NT) dstset[i] = (i < nl * N) ? src[i] : make_float2(0.f, 0.f);
        __syncthreads();
        line_fft<N, LK, B_NT, N, false>(dstset, tw, tid);
    } else if (MODE == B_SOLVE_INV) {
        const float2* src = a.src + (size_t)item * a.src_stride + line0;
        for (int i = tid; i < LK * N; i += B_NT) {
            const bool ok = i < nl * N;
            set0[i] = ok ? src[i] : make_float2(0.f, 0.f);
            set1[i] = ok ? src[a.in_plane_stride + i] : make_float2(0.f, 0.f);
        }
        __syncthreads();
        line_fft<N, 2 * LK, B_NT, N, false>(set0, tw, tid);
    }

    if (MODE == B_INV) {
        const float2* src = a.src + (size_t)item * a.src_stride + line0;
        for (int i = tid; i < LK * N; i += B_NT) set0[i] = (i < nl * N) ? src[i] : make_float2(0.f, 0.f);
        __syncthreads();
        line_fft<N, LK, B_NT, N, true>(set0, tw, tid);
        float2* d = a.dst + (size_t)item * a.dst_stride + line0;
        for (int i = tid; i < nl * N; i += B_NT) d[i] = set0[i];
        return;
    }

    // ---- stage 2: pointwise
    if (MODE == B_FWD) {
        float2* dst = a.dst + (size_t)(a.dst_slot ? a.dst_slot[item] : item) * a.dst_stride + line0;
        for (int i = tid; i < nl * N; i += B_NT) dst[i] = set0[i];
        return;
    }
    if (MODE == B_FWD_ABS_INV) {
        // fft_result = FFT(image);  IFFT(fft_result.abs())   (correlation_flow.cc:91-92)
        float2* dst = a.dst + (size_t)(a.dst_slot ? a.dst_slot[item] : item) * a.dst_stride + line0;
        for (int i = tid; i < LK * N; i += B_NT) {
            const float2 f = set0[i];
            if (i < nl * N) dst[i] = f;
            set0[i] = make_float2(sqrtf(f.x * f.x + f.y * f.y), 0.f);
        }
        __syncthreads();
        line_fft<N, LK, B_NT, N, true>(set0, tw, tid);
        float2* d2 = a.dst2 + (size_t)item * a.dst2_stride + line0;
        for (int i = tid; i < nl * N; i += B_NT) d2[i] = set0[i];
        return;
    }
    if (MODE == B_MUL_INV || MODE == B_FWD_MUL_INV) {
        // xzf = xf * zf.conjugate() for (z,z) and (x,z)   (correlation_flow.cc:210-211,220-221)
        const float2* z = a.zsrc + (size_t)(a.z_idx ? a.z_idx[item] : item) * a.z_stride + line0;
        const float2* x = nullptr;
        if (MODE == B_MUL_INV) x = a.src + (size_t)(a.src_idx ? a.src_idx[item] : item) * a.src_stride + line0;
        for (int i = tid; i < LK * N; i += B_NT) {
            float2 zz = make_float2(0.f, 0.f), xz = make_float2(0.f, 0.f);
            if (i < nl * N) {
                const float2 zv = z[i];
                const float2 xv = (MODE == B_MUL_INV) ? x[i] : set1[i];
                zz = make_float2(zv.x * zv.x + zv.y * zv.y, 0.f);
                xz = cmulc(xv, zv);
            }
            set0[i] = zz; set1[i] = xz;
        }
        __syncthreads();
        line_fft<N, 2 * LK, B_NT, N, true>(set0, tw, tid);
        float2* d = a.dst + (size_t)item * a.dst_stride + line0;
        for (int i = tid; i < nl * N; i += B_NT) { d[i] = set0[i]; d[a.out_plane_stride + i] = set1[i]; }
        return;
    }
    if (MODE == B_SOLVE_INV) {
        // H = T/(Kzz + lambda); G = H * Kxz   (correlation_flow.cc:171-172), T[k][l] = (-1)^(k+l)
        const float rzz = 1.f / __uint_as_float(a.maxbuf[2 * item + 0]);
        const float rxz = 1.f / __uint_as_float(a.maxbuf[2 * item + 1]);
        for (int i = tid; i < LK * N; i += B_NT) {
            const int line = i / N, l = i - line * N;
            const float2 kzz = set0[i], kxz = set1[i];
            const float2 den = make_float2(kzz.x * rzz + a.lambda, kzz.y * rzz);
            const float2 num = make_float2(kxz.x * rxz, kxz.y * rxz);
            const float inv = 1.f / (den.x * den.x + den.y * den.y);
            float2 g = cmulc(num, den);
            const float sgn = ((k0 + line + l) & 1) ? -inv : inv;
            g.x *= sgn; g.y *= sgn;
            if (!(i < nl * N)) g = make_float2(0.f, 0.f);
            set0[i] = g;
        }
        __syncthreads();
        line_fft<N, LK, B_NT, N, true>(set0, tw, tid);
        float2* d = a.dst + (size_t)item * a.dst_stride + line0;
        for (int i = tid; i < nl * N; i += B_NT) d[i] = set0[i];
        return;
    }
}

template <int N, int MODE> static void launchB_t(hipStream_t s, int n_items, const BArgs& a) {
    constexpr int LK = BCfg<N>::LK;
    dim3 grid((a.hr + LK - 1) / LK, n_items), block(B_NT);
    hipLaunchKernelGGL((kB<N, MODE>), grid, block, BCfg<N>::BYTES, s, a);
}

#define DISPATCH_LINE(n, CALL)             \
    switch (n) {                           \
        case 80:   { CALL(80);   break; }  \
        case 160:  { CALL(160);  break; }  \
        case 320:  { CALL(320);  break; }  \
        case 480:  { CALL(480);  break; }  \
        case 640:  { CALL(640);  break; }  \
        case 1280: { CALL(1280); break; }  \
        default: break;                    \
    }

static BArgs base_bargs(PlaneGeom g, Tables t) {
    BArgs a{};
    a.cols = g.cols; a.hr = g.hr; a.tw_cols = t.tw_cols;
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
                          float2* dstF_base, size_t dstF_stride, const int* dst_slot, float2* tmp, size_t tmp_stride) {
    BArgs a = base_bargs(g, t);
    a.src = src; a.src_stride = src_stride; a.dst = dstF_base; a.dst_stride = dstF_stride; a.dst_slot = dst_slot;
    a.dst2 = tmp; a.dst2_stride = tmp_stride;
#define CALL(N) launchB_t<N, B_FWD_ABS_INV>(s, n_items, a)
    DISPATCH_LINE(g.cols, CALL)
#undef CALL
}
void launch_B_mul_inv(hipStream_t s, int n_items, PlaneGeom g, Tables t, bool x_fwd,
                      const float2* xsrc, size_t x_stride, const int* x_idx,
                      const float2* zsrc, size_t z_stride, const int* z_idx,
                      float2* out, size_t item_stride, size_t plane_stride) {
    BArgs a = base_bargs(g, t);
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
void launch_B_solve_inv(hipStream_t s, int n_items, PlaneGeom g, Tables t, const float2* buf, size_t item_stride,
                        size_t plane_stride, const unsigned* maxbuf, float lambda, float2* out, size_t out_stride) {
    BArgs a = base_bargs(g, t);
    a.src = buf; a.src_stride = item_stride; a.in_plane_stride = plane_stride; a.maxbuf = maxbuf; a.lambda = lambda;
    a.dst = out; a.dst_stride = out_stride;
#define CALL(N) launchB_t<N, B_SOLVE_INV>(s, n_items, a)
    DISPATCH_LINE(g.cols, CALL)
#undef CALL
}

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
__global__ void k_energy(const float2* __restrict__ xsrc, size_t x_stride, const int* x_idx,
                         const float2* __restrict__ zsrc, size_t z_stride, const int* z_idx, size_t n, float* energy) {
    __shared__ double red[256];
    const int item = blockIdx.x, which = blockIdx.y, tid = threadIdx.x;
    const float2* p = which == 0 ? xsrc + (size_t)(x_idx ? x_idx[item] : item) * x_stride
                                 : zsrc + (size_t)(z_idx ? z_idx[item] : item) * z_stride;
    double acc = 0.0;
    for (size_t i = tid; i < n; i += 256) { const float2 v = p[i]; acc += (double)(v.x * v.x + v.y * v.y); }
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    if (tid == 0) energy[2 * item + which] = (float)red[0];
}
void launch_energy(hipStream_t s, int n_items, PlaneGeom g, const float2* xsrc, size_t x_stride, const int* x_idx,
                   const float2* zsrc, size_t z_stride, const int* z_idx, float* energy) {
    hipLaunchKernelGGL(k_energy, dim3(n_items, 2), dim3(256), 0, s, xsrc, x_stride, x_idx, zsrc, z_stride, z_idx,
                       (size_t)g.hr * g.cols, energy);
}

__global__ void k_finalize(const Partial* __restrict__ partials, int partial_stride, int n_partials, SurfaceResult* out) {
    const int item = blockIdx.x;
    if (threadIdx.x != 0) return;
    const Partial* p = partials + (size_t)item * partial_stride;
    SurfaceResult r; r.sum = 0; r.sumsq = 0; r.peak = -INFINITY; r.idx = 0x7FFFFFFF;
    for (int i = 0; i < n_partials; ++i) {           // fixed order -> deterministic
        r.sum += p[i].sum; r.sumsq += p[i].sumsq;
        if (p[i].peak > r.peak || (p[i].peak == r.peak && p[i].idx < r.idx)) { r.peak = p[i].peak; r.idx = p[i].idx; }
    }
    out[item] = r;
}
void launch_finalize(hipStream_t s, int n_items, const Partial* partials, int partial_stride, int n_partials, SurfaceResult* out) {
    hipLaunchKernelGGL(k_finalize, dim3(n_items), dim3(64), 0, s, partials, partial_stride, n_partials, out);
}

__global__ void k_rot_index(int n_items, const SurfaceResult* rot_res, const int* pair, const int* variant, int PD, int* rot_index) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_items) rot_index[t] = variant[t] * PD + (rot_res[pair[t]].idx % PD);
}
void launch_rot_index(hipStream_t s, int n_items, const SurfaceResult* rot_res, const int* pair, const int* variant, int PD, int* rot_index) {
    hipLaunchKernelGGL(k_rot_index, dim3((n_items + 63) / 64), dim3(64), 0, s, n_items, rot_res, pair, variant, PD, rot_index);
}

__global__ void k_dbg_rot(const float* __restrict__ img, RotEntry R, float* __restrict__ out, int H, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < H * W) out[i] = rot_sample(img, H, W, R, i % H, i / H);
}
void launch_dbg_rot(hipStream_t s, const float* img, const RotEntry& R, float* out, int H, int W) {
    hipLaunchKernelGGL(k_dbg_rot, dim3((H * W + 255) / 256), dim3(256), 0, s, img, R, out, H, W);
}
__global__ void k_dbg_polar(const float* __restrict__ p, const uint32_t* __restrict__ tab, float* __restrict__ out,
                            int H, int W, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // out is column-major PD x PC == tab order [PC][PD]
    if (i < n) out[i] = polar_sample(p, H, W, tab[i]);
}
void launch_dbg_polar(hipStream_t s, const float* p, const uint32_t* tab, float* out, int H, int W, int PD, int PC) {
    hipLaunchKernelGGL(k_dbg_polar, dim3((PD * PC + 255) / 256), dim3(256), 0, s, p, tab, out, H, W, PD * PC);
}

__global__ void k_transpose_c(const float2* __restrict__ src, float2* __restrict__ dst, int R, int C) {
    __shared__ float2 tile[32][33];
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
    hipLaunchKernelGGL(k_transpose_c, grid, block, 0, s, src, dst, src_rows, src_cols);
}

}  // namespace kcc
