/*
 * kcc_oracle.c -- CPU restatement (parity ORACLE) of NI-SLAM's KCC front end.
 * TEST INFRASTRUCTURE ONLY -- see kcc_oracle.h for the rules and the
 * "PARITY UNPINNED" statement.  Every function cites the reference lines it
 * follows (paths relative to /root/reference).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; no -ffast-math, so
 * float expressions evaluate exactly as written).
 */
#include "kcc_oracle.h"

#include <malloc.h>
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif


/* ------------------------------------------------------------------ */
/* per-thread stack arena for the plane-sized temporaries              */
/* ------------------------------------------------------------------ */
/* The reference allocates its Eigen temporaries per call; so did this port -- and on a 256-thread host the allocator
 * (mmap / page faults / arena trimming behind malloc) stopped the OpenMP batch driver from scaling beyond 16 threads
 * (profiles/r04_cpu_scale.txt).  ora_track_pairs gives every thread ONE block up front; the temporaries of the hot path
 * are carved from it stack-wise (every function frees what it allocated before it returns).  Outside ora_track_pairs no
 * arena is reserved and tmp_alloc IS malloc.  Timing infrastructure only: no arithmetic changes. */
static __thread char* t_arena = NULL; static __thread size_t t_cap = 0, t_top = 0;
static void* tmp_alloc(size_t n) {
    n = (n + 63) & ~(size_t)63;
    if (t_arena && t_top + n <= t_cap) { void* p = t_arena + t_top; t_top += n; return p; }
    return malloc(n);
}
static void tmp_free(void* p) {
    if (!p) return;
    if (t_arena && (char*)p >= t_arena && (char*)p < t_arena + t_cap) { const size_t off = (size_t)((char*)p - t_arena); if (off < t_top) t_top = off; return; }
    free(p);
}
static void tmp_reserve(size_t bytes) { t_arena = (char*)malloc(bytes); t_cap = t_arena ? bytes : 0; t_top = 0; if (t_arena) memset(t_arena, 0, bytes); }
static void tmp_release(void) { free(t_arena); t_arena = NULL; t_cap = t_top = 0; }

/* ------------------------------------------------------------------ */
/* float32 mixed-radix FFT (stands in for FFTW3f, which is un-vendored) */
/* ------------------------------------------------------------------ */

#define ORA_MAX_PLANS 16
#define ORA_MAX_FAC   32

typedef struct {
    int n;
    int nf;
    int fac[ORA_MAX_FAC];
    ora_cf32* tw;       /* tw[t] = exp(-2*pi*i*t/n), double-evaluated, rounded to float */
} ora_plan;

struct ora_ctx {
    ora_config cfg;
    int H, W, PD, PC;
    ora_plan plans[ORA_MAX_PLANS];
    int nplans;
    ora_cf32* target_fft;           /* (H/2+1) x W   correlation_flow.cc:42 */
    ora_cf32* target_rotation_fft;  /* (PD/2+1) x PC correlation_flow.cc:43 */
    /* warpPolar maps (depend only on H, W, PD, PC) */
    float* mapx; float* mapy;
    /* coarse-to-fine extension (BASELINE config 3, no reference counterpart): arg-max windows, [0] translation
       surface, [1] rotation surface; radius < 0 = off */
    int win_row[2], win_col[2], win_radius;
    int force_rot_row, force_rot_col;   /* test hook: imposed rotation arg-max (row < 0: off) */
};

static inline ora_cf32 cmul(ora_cf32 a, ora_cf32 b) {
    ora_cf32 r; r.re = a.re * b.re - a.im * b.im; r.im = a.re * b.im + a.im * b.re; return r;
}
static inline ora_cf32 cadd(ora_cf32 a, ora_cf32 b) { ora_cf32 r = { a.re + b.re, a.im + b.im }; return r; }
static inline ora_cf32 csub(ora_cf32 a, ora_cf32 b) { ora_cf32 r = { a.re - b.re, a.im - b.im }; return r; }
static inline ora_cf32 cconj(ora_cf32 a) { ora_cf32 r = { a.re, -a.im }; return r; }

static void plan_init(ora_plan* p, int n) {
    static const int radices[] = { 4, 2, 3, 5, 7 };
    p->n = n; p->nf = 0;
    int m = n;
    for (int i = 0; i < 5; ++i)
        while (m % radices[i] == 0 && p->nf < ORA_MAX_FAC) { p->fac[p->nf++] = radices[i]; m /= radices[i]; }
    for (int f = 11; m > 1; f += 2)      /* generic odd primes, direct DFT butterflies */
        while (m % f == 0 && p->nf < ORA_MAX_FAC) { p->fac[p->nf++] = f; m /= f; }
    p->tw = (ora_cf32*)malloc(sizeof(ora_cf32) * (size_t)n);
    for (int t = 0; t < n; ++t) {
        double a = -2.0 * M_PI * (double)t / (double)n;
        p->tw[t].re = (float)cos(a); p->tw[t].im = (float)sin(a);
    }
}

/* a context is single-owner (not thread-safe), like the reference's CorrelationFlow: no locking */
static const ora_plan* get_plan(ora_ctx* ctx, int n) {
    for (int i = 0; i < ctx->nplans; ++i) if (ctx->plans[i].n == n) return &ctx->plans[i];
    if (ctx->nplans >= ORA_MAX_PLANS) return NULL;
    plan_init(&ctx->plans[ctx->nplans], n);
    return &ctx->plans[ctx->nplans++];
}

/* One Stockham autosort pass of radix r (Ns = product of the radices already applied). */
static void stockham_pass(const ora_cf32* in, ora_cf32* out, int n, int r, int Ns,
                          const ora_cf32* tw, int inverse) {
    const int m = n / r;
    const int tstep = n / (Ns * r);
    const int rstep = n / r;              /* W_r^t = tw[t*rstep] */
    ora_cf32 vs[64], ys[64];
    ora_cf32* v = vs; ora_cf32* y = ys;
    if (r > 64) { v = (ora_cf32*)tmp_alloc(sizeof(ora_cf32) * 2 * (size_t)r); y = v + r; }   /* a prime factor beyond 61 (e.g. 158 = 2 x 79) */
    for (int j = 0; j < m; ++j) {
        const int k = j % Ns;
        for (int q = 0; q < r; ++q) {
            ora_cf32 a = in[j + q * m];
            if (q && k) {
                ora_cf32 w = tw[q * k * tstep];
                if (inverse) w.im = -w.im;
                a = cmul(a, w);
            }
            v[q] = a;
        }
        if (r == 2) {
            y[0] = cadd(v[0], v[1]); y[1] = csub(v[0], v[1]);
        } else if (r == 4) {
            ora_cf32 a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]);
            ora_cf32 b0 = cadd(v[1], v[3]), b1 = csub(v[1], v[3]);
            /* forward: -i*b1 ; inverse: +i*b1 */
            ora_cf32 ib1;
            if (!inverse) { ib1.re = b1.im; ib1.im = -b1.re; } else { ib1.re = -b1.im; ib1.im = b1.re; }
            y[0] = cadd(a0, b0); y[2] = csub(a0, b0);
            y[1] = cadd(a1, ib1); y[3] = csub(a1, ib1);
        } else {
            for (int q = 0; q < r; ++q) {
                ora_cf32 acc = v[0];
                for (int s = 1; s < r; ++s) {
                    ora_cf32 w = tw[((q * s) % r) * rstep];
                    if (inverse) w.im = -w.im;
                    acc = cadd(acc, cmul(v[s], w));
                }
                y[q] = acc;
            }
        }
        const int j0 = (j / Ns) * Ns * r + k;
        for (int q = 0; q < r; ++q) out[j0 + q * Ns] = y[q];
    }
    if (v != vs) tmp_free(v);
}

/* In-place unnormalised complex FFT of contiguous data[n]; work[n] scratch. */
static void cfft(const ora_plan* p, ora_cf32* data, ora_cf32* work, int inverse) {
    ora_cf32* a = data; ora_cf32* b = work;
    int Ns = 1;
    for (int i = 0; i < p->nf; ++i) {
        stockham_pass(a, b, p->n, p->fac[i], Ns, p->tw, inverse);
        Ns *= p->fac[i];
        ora_cf32* t = a; a = b; b = t;
    }
    if (a != data) memcpy(data, a, sizeof(ora_cf32) * (size_t)p->n);
}

/* real line of n (even) floats -> n/2+1 complex, via one complex FFT of length n/2 */
static void rfft_line(const ora_plan* ph, const ora_plan* pn, const float* x, ora_cf32* X,
                      ora_cf32* z, ora_cf32* work) {
    const int h = ph->n;
    for (int m = 0; m < h; ++m) { z[m].re = x[2 * m]; z[m].im = x[2 * m + 1]; }
    cfft(ph, z, work, 0);
    X[0].re = z[0].re + z[0].im; X[0].im = 0.f;
    X[h].re = z[0].re - z[0].im; X[h].im = 0.f;
    for (int k = 1; k < h; ++k) {
        ora_cf32 A = z[k], B = cconj(z[h - k]);
        ora_cf32 E = { 0.5f * (A.re + B.re), 0.5f * (A.im + B.im) };
        ora_cf32 D = { 0.5f * (A.re - B.re), 0.5f * (A.im - B.im) };
        ora_cf32 O = { D.im, -D.re };                /* -i * D */
        X[k] = cadd(E, cmul(pn->tw[k], O));
    }
}

/* n/2+1 complex (Hermitian half, imag of X[0], X[n/2] ignored like FFTW c2r) -> n reals, unnormalised */
static void irfft_line(const ora_plan* ph, const ora_plan* pn, const ora_cf32* X, float* x,
                       ora_cf32* z, ora_cf32* work) {
    const int h = ph->n;
    z[0].re = X[0].re + X[h].re; z[0].im = X[0].re - X[h].re;
    for (int k = 1; k < h; ++k) {
        ora_cf32 A = X[k], B = cconj(X[h - k]);
        ora_cf32 S = cadd(A, B), D = csub(A, B);
        ora_cf32 wD = cmul(cconj(pn->tw[k]), D);
        ora_cf32 iwD = { -wD.im, wD.re };             /* +i * wD */
        z[k] = cadd(S, iwD);
    }
    cfft(ph, z, work, 1);
    for (int m = 0; m < h; ++m) { x[2 * m] = z[m].re; x[2 * m + 1] = z[m].im; }
}

/* CorrelationFlow::FFT  correlation_flow.cc:53-63.
 * fftwf_plan_dft_r2c_2d(n0 = cols, n1 = rows): the column-major rows x cols array is a
 * row-major cols x rows array to FFTW, so the halved axis is the row axis; unnormalised. */
void ora_fft(ora_ctx* ctx, const float* x, int rows, int cols, ora_cf32* xf) {
    const int h = rows / 2, hr = h + 1;
    const ora_plan* ph = get_plan(ctx, h);
    const ora_plan* pn = get_plan(ctx, rows);
    const ora_plan* pc = get_plan(ctx, cols);
    int mx = rows > cols ? rows : cols;
    ora_cf32* z = (ora_cf32*)tmp_alloc(sizeof(ora_cf32) * (size_t)mx * 2);
    ora_cf32* work = z + mx;
    for (int c = 0; c < cols; ++c) rfft_line(ph, pn, x + (size_t)c * rows, xf + (size_t)c * hr, z, work);
    for (int k = 0; k < hr; ++k) {
        for (int c = 0; c < cols; ++c) z[c] = xf[(size_t)c * hr + k];
        cfft(pc, z, work, 0);
        for (int c = 0; c < cols; ++c) xf[(size_t)c * hr + k] = z[c];
    }
    tmp_free(z);
}

/* CorrelationFlow::IFFT  correlation_flow.cc:65-77: c2r then x / x.size(). Requires even rows. */
void ora_ifft(ora_ctx* ctx, const ora_cf32* xf, int hrows, int cols, float* x) {
    const int rows = (hrows - 1) * 2, h = rows / 2, hr = hrows;
    const ora_plan* ph = get_plan(ctx, h);
    const ora_plan* pn = get_plan(ctx, rows);
    const ora_plan* pc = get_plan(ctx, cols);
    int mx = rows > cols ? rows : cols;
    ora_cf32* cxf = (ora_cf32*)tmp_alloc(sizeof(ora_cf32) * (size_t)hr * cols);   /* :68 copy */
    memcpy(cxf, xf, sizeof(ora_cf32) * (size_t)hr * cols);
    ora_cf32* z = (ora_cf32*)tmp_alloc(sizeof(ora_cf32) * (size_t)mx * 2);
    ora_cf32* work = z + mx;
    for (int k = 0; k < hr; ++k) {
        for (int c = 0; c < cols; ++c) z[c] = cxf[(size_t)c * hr + k];
        cfft(pc, z, work, 1);
        for (int c = 0; c < cols; ++c) cxf[(size_t)c * hr + k] = z[c];
    }
    for (int c = 0; c < cols; ++c) irfft_line(ph, pn, cxf + (size_t)c * hr, x + (size_t)c * rows, z, work);
    const float size = (float)((long)rows * cols);                              /* :76 x/x.size() */
    for (long i = 0; i < (long)rows * cols; ++i) x[i] = x[i] / size;
    tmp_free(z); tmp_free(cxf);
}

/* ------------------------------------------------------------------ */
/* OpenCV 4.2 imgproc restatements (un-vendored third party; [recalled]) */
/* ------------------------------------------------------------------ */

enum { INTER_BITS = 5, INTER_TAB_SIZE = 32, AB_BITS = 10, AB_SCALE = 1024 };
enum { BORDER_CONSTANT0 = 0, BORDER_WRAP_ = 3 };

static inline int cv_round_f(float v)  { return (int)lrintf(v); }  /* round-half-even */
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

/* borderInterpolate(p, len, BORDER_WRAP) */
static inline int wrap_idx(int p, int len) {
    if (p < 0) p -= ((p - len + 1) / len) * len;
    if (p >= len) p %= len;
    return p;
}

/* remapBilinear<float> for one destination pixel.  src is the Eigen column-major array
 * standing for the row-major cv::Mat (Mat(y,x) == arr[x*rows + y]);  fx,fy in [0,32). */
static inline float remap_bilinear_px(const float* src, int rows, int cols, int sx, int sy,
                                      int fx, int fy, int border) {
    const float scale = 1.f / INTER_TAB_SIZE;               /* initInterTab1D */
    const float tx0 = 1.f - fx * scale, tx1 = fx * scale;   /* interpolateLinear */
    const float ty0 = 1.f - fy * scale, ty1 = fy * scale;
    const float w0 = ty0 * tx0, w1 = ty0 * tx1, w2 = ty1 * tx0, w3 = ty1 * tx1;  /* initInterTab2D */
    const int width1 = cols - 1 > 0 ? cols - 1 : 0, height1 = rows - 1 > 0 ? rows - 1 : 0;
    float v0, v1, v2, v3;
    if ((unsigned)sx < (unsigned)width1 && (unsigned)sy < (unsigned)height1) {
        v0 = src[(size_t)sx * rows + sy];       v1 = src[(size_t)(sx + 1) * rows + sy];
        v2 = src[(size_t)sx * rows + sy + 1];   v3 = src[(size_t)(sx + 1) * rows + sy + 1];
    } else if (border == BORDER_CONSTANT0) {
        if (sx >= cols || sx + 1 < 0 || sy >= rows || sy + 1 < 0) return 0.f;
        const int x0 = sx, x1 = sx + 1, y0 = sy, y1 = sy + 1;
        v0 = (x0 >= 0 && x0 < cols && y0 >= 0 && y0 < rows) ? src[(size_t)x0 * rows + y0] : 0.f;
        v1 = (x1 >= 0 && x1 < cols && y0 >= 0 && y0 < rows) ? src[(size_t)x1 * rows + y0] : 0.f;
        v2 = (x0 >= 0 && x0 < cols && y1 >= 0 && y1 < rows) ? src[(size_t)x0 * rows + y1] : 0.f;
        v3 = (x1 >= 0 && x1 < cols && y1 >= 0 && y1 < rows) ? src[(size_t)x1 * rows + y1] : 0.f;
    } else {
        const int x0 = wrap_idx(sx, cols), x1 = wrap_idx(sx + 1, cols);
        const int y0 = wrap_idx(sy, rows), y1 = wrap_idx(sy + 1, rows);
        v0 = src[(size_t)x0 * rows + y0]; v1 = src[(size_t)x1 * rows + y0];
        v2 = src[(size_t)x0 * rows + y1]; v3 = src[(size_t)x1 * rows + y1];
    }
    return v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
}

/* cv::warpPolar map construction (linear, forward map) for dsize = (PC cols, PD rows),
 * centre = ((float)W/2, (float)H/2), maxRadius = min(H/2, W/2)   correlation_flow.cc:231-234 */
static void build_polar_maps(ora_ctx* ctx) {
    const int PD = ctx->PD, PC = ctx->PC;
    ctx->mapx = (float*)malloc(sizeof(float) * (size_t)PD * PC);
    ctx->mapy = (float*)malloc(sizeof(float) * (size_t)PD * PC);
    const float cx = (float)ctx->W / 2, cy = (float)ctx->H / 2;
    const int rh = ctx->H / 2, rw = ctx->W / 2;
    const double maxRadius = (double)(rh < rw ? rh : rw);
    const double Kangle = (2.0 * 3.1415926535897932384626433832795) / PD;       /* CV_2PI / dsize.height */
    const double Kmag = maxRadius / PC;
    float* rhos = (float*)malloc(sizeof(float) * (size_t)PC);
    for (int rho = 0; rho < PC; ++rho) rhos[rho] = (float)(rho * Kmag);
    for (int phi = 0; phi < PD; ++phi) {
        const double KKy = Kangle * phi;
        const double cp = cos(KKy), sp = sin(KKy);
        for (int rho = 0; rho < PC; ++rho) {
            const double x = rhos[rho] * cp + cx;
            const double y = rhos[rho] * sp + cy;
            ctx->mapx[(size_t)phi * PC + rho] = (float)x;
            ctx->mapy[(size_t)phi * PC + rho] = (float)y;
        }
    }
    free(rhos);
}

/* CorrelationFlow::polar  correlation_flow.cc:228-236:
 * warpPolar(INTER_LINEAR | WARP_FILL_OUTLIERS) == remap(float maps, INTER_LINEAR, BORDER_CONSTANT 0);
 * output Mat PD x PC converted back to a column-major PD x PC array. */
void ora_polar(ora_ctx* ctx, const float* x, float* out) {
    const int PD = ctx->PD, PC = ctx->PC;
    for (int phi = 0; phi < PD; ++phi)
        for (int rho = 0; rho < PC; ++rho) {
            const int sxq = cv_round_f(ctx->mapx[(size_t)phi * PC + rho] * INTER_TAB_SIZE);
            const int syq = cv_round_f(ctx->mapy[(size_t)phi * PC + rho] * INTER_TAB_SIZE);
            const int sx = sat_short(sxq >> INTER_BITS), sy = sat_short(syq >> INTER_BITS);
            out[(size_t)rho * PD + phi] = remap_bilinear_px(x, ctx->H, ctx->W, sx, sy,
                sxq & (INTER_TAB_SIZE - 1), syq & (INTER_TAB_SIZE - 1), BORDER_CONSTANT0);
        }
}

/* cv::warpAffine(INTER_LINEAR, BORDER_WRAP) with a forward 2x3 matrix M (double). */
static void warp_affine_wrap(const float* x, int rows, int cols, const double Min[6], float* out) {
    double M[6]; memcpy(M, Min, sizeof(M));
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    double b1 = -M[0] * M[2] - M[1] * M[5];
    double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    const int round_delta = AB_SCALE / INTER_TAB_SIZE / 2;
    int* adelta = (int*)tmp_alloc(sizeof(int) * (size_t)cols * 2);
    int* bdelta = adelta + cols;
    for (int c = 0; c < cols; ++c) {
        adelta[c] = cv_round_d(M[0] * c * AB_SCALE);
        bdelta[c] = cv_round_d(M[3] * c * AB_SCALE);
    }
    for (int r = 0; r < rows; ++r) {
        const int X0 = cv_round_d((M[1] * r + M[2]) * AB_SCALE) + round_delta;
        const int Y0 = cv_round_d((M[4] * r + M[5]) * AB_SCALE) + round_delta;
        for (int c = 0; c < cols; ++c) {
            const int X = (X0 + adelta[c]) >> (AB_BITS - INTER_BITS);
            const int Y = (Y0 + bdelta[c]) >> (AB_BITS - INTER_BITS);
            const int sx = sat_short(X >> INTER_BITS), sy = sat_short(Y >> INTER_BITS);
            out[(size_t)c * rows + r] = remap_bilinear_px(x, rows, cols, sx, sy,
                X & (INTER_TAB_SIZE - 1), Y & (INTER_TAB_SIZE - 1), BORDER_WRAP_);
        }
    }
    tmp_free(adelta);
}

/* cv::getRotationMatrix2D(center (Point2f), angle [deg], scale = 1) */
static void rotation_matrix_2d(float cx, float cy, double angle, double m[6]) {
    angle *= 3.1415926535897932384626433832795 / 180;
    const double alpha = cos(angle) * 1.0, beta = sin(angle) * 1.0;
    m[0] = alpha; m[1] = beta;  m[2] = (1 - alpha) * cx - beta * cy;
    m[3] = -beta; m[4] = alpha; m[5] = beta * cx + (1 - alpha) * cy;
}

/* RotateArray  utils.cc:154-161 : centre (cols/2., rows/2.) as Point2f */
void ora_rotate(const float* x, int rows, int cols, float degree, float* out) {
    double m[6];
    rotation_matrix_2d((float)(cols / 2.), (float)(rows / 2.), (double)degree, m);
    warp_affine_wrap(x, rows, cols, m, out);
}

/* WarpArray  utils.cc:163-171 (only reached by the dead `rectify` of correlation_flow.cc:141) */
void ora_warp(const float* x, int rows, int cols, float tx, float ty, float degree, float* out) {
    const double m[6] = { 1, 0, (double)tx, 0, 1, (double)ty };  /* CV_32F matrix converted to CV_64F */
    float* tmp = (float*)tmp_alloc(sizeof(float) * (size_t)rows * cols);
    warp_affine_wrap(x, rows, cols, m, tmp);
    ora_rotate(tmp, rows, cols, degree, out);
    tmp_free(tmp);
}

/* NormalizeDegree  utils.cc:173-175 */
double ora_normalize_degree(double a) { return a - 360 * floor((a + 180) / 360); }

/* ConvertMatToNormalizedArray  utils.cc:110-118 : cv2eigen (u8 -> f32, transpose) then /255.0 */
void ora_normalize_u8(const uint8_t* img, int rows, int cols, float* out) {
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c)
            out[(size_t)c * rows + r] = (float)img[(size_t)r * cols + c] / 255.0f;
}

/* ------------------------------------------------------------------ */
/* CorrelationFlow                                                      */
/* ------------------------------------------------------------------ */

/* GetTargetFFT  correlation_flow.cc:46-51 */
static ora_cf32* get_target_fft(ora_ctx* ctx, int rows, int cols) {
    float* t = (float*)calloc((size_t)rows * cols, sizeof(float));
    t[(size_t)(cols / 2) * rows + rows / 2] = 1.f;
    ora_cf32* f = (ora_cf32*)malloc(sizeof(ora_cf32) * (size_t)(rows / 2 + 1) * cols);
    ora_fft(ctx, t, rows, cols, f);
    free(t);
    return f;
}

/* ctor  correlation_flow.cc:37-44 : cfg.height/width overridden by the camera's image size */
ora_ctx* ora_create(const ora_config* cfg, int image_height, int image_width) {
    if (!cfg || image_height <= 0 || image_width <= 0 || (image_height & 1) || (cfg->rotation_divisor & 1)) return NULL;
    ora_ctx* ctx = (ora_ctx*)calloc(1, sizeof(ora_ctx));
    ctx->cfg = *cfg;
    ctx->cfg.height = image_height; ctx->cfg.width = image_width;
    ctx->H = image_height; ctx->W = image_width;
    ctx->PD = cfg->rotation_divisor; ctx->PC = cfg->rotation_channel;
    ctx->win_radius = -1; ctx->force_rot_row = -1; ctx->force_rot_col = -1;
    ctx->target_fft = get_target_fft(ctx, ctx->H, ctx->W);
    ctx->target_rotation_fft = get_target_fft(ctx, ctx->PD, ctx->PC);
    build_polar_maps(ctx);
    return ctx;
}

void ora_destroy(ora_ctx* ctx) {
    if (!ctx) return;
    for (int i = 0; i < ctx->nplans; ++i) free(ctx->plans[i].tw);
    free(ctx->target_fft); free(ctx->target_rotation_fft); free(ctx->mapx); free(ctx->mapy);
    free(ctx);
}
int ora_rows(const ora_ctx* c) { return c->H; }
int ora_cols(const ora_ctx* c) { return c->W; }

/* RemoveZeroComponent  correlation_flow.cc:79-87.  Both block assignments read the ORIGINAL x,
 * so y(0,0) = (x(0,1)+x(0,cols-1))/2 (second statement wins).  "/2.0" on a float array is a
 * float division in Eigen (the scalar is cast to float). */
void ora_remove_zero(const float* x, int rows, int cols, float* y) {
    memcpy(y, x, sizeof(float) * (size_t)rows * cols);
    for (int c = 0; c < cols; ++c)
        y[(size_t)c * rows + 0] = (x[(size_t)c * rows + 1] + x[(size_t)c * rows + rows - 1]) / 2.0f;
    for (int r = 0; r < rows; ++r)
        y[r] = (x[(size_t)1 * rows + r] + x[(size_t)(cols - 1) * rows + r]) / 2.0f;
}

/* fftshift  circ_shift.h:238-244 with the index map of :131-154:
 * out(r,c) = in((r - rows/2) mod rows, (c - cols/2) mod cols) */
void ora_fftshift(const float* x, int rows, int cols, float* y) {
    const int rs = rows / 2, cs = cols / 2;
    for (int c = 0; c < cols; ++c) {
        int sc = c - cs; if (sc >= cols) sc -= cols; if (sc < 0) sc += cols;
        for (int r = 0; r < rows; ++r) {
            int sr = r - rs; if (sr >= rows) sr -= rows; if (sr < 0) sr += rows;
            y[(size_t)c * rows + r] = x[(size_t)sc * rows + sr];
        }
    }
}

/* ComputeIntermedium  correlation_flow.cc:89-95 */
void ora_intermedium(ora_ctx* ctx, const float* image, ora_cf32* fft_result, ora_cf32* fft_polar) {
    const int H = ctx->H, W = ctx->W, hr = H / 2 + 1;
    const size_t n = (size_t)H * W, nc = (size_t)hr * W;
    ora_fft(ctx, image, H, W, fft_result);                                  /* :91 */
    ora_cf32* mag = (ora_cf32*)tmp_alloc(sizeof(ora_cf32) * nc);
    for (size_t i = 0; i < nc; ++i) {                                       /* :92 fft_result.abs() */
        mag[i].re = hypotf(fft_result[i].re, fft_result[i].im); mag[i].im = 0.f;
    }
    float* power = (float*)tmp_alloc(sizeof(float) * n * 3);
    float* high = power + n; float* shifted = high + n;
    ora_ifft(ctx, mag, hr, W, power);
    ora_remove_zero(power, H, W, high);                                     /* :93 */
    ora_fftshift(high, H, W, shifted);                                      /* :94 */
    float* pol = (float*)tmp_alloc(sizeof(float) * (size_t)ctx->PD * ctx->PC);
    ora_polar(ctx, shifted, pol);
    ora_fft(ctx, pol, ctx->PD, ctx->PC, fft_polar);
    tmp_free(pol); tmp_free(power); tmp_free(mag);
}

/* Eigen-like reduction order: 16 interleaved float accumulators, then a pairwise fold
 * (a stand-in for Eigen's packet reduction; the exact packet width depends on -march=native). */
static float sum16(const float* v, long n) {
    float acc[16] = { 0 };
    long i = 0;
    for (; i + 16 <= n; i += 16) for (int l = 0; l < 16; ++l) acc[l] += v[i + l];
    for (int l = 0; i < n; ++i, ++l) acc[l] += v[i];
    for (int s = 8; s >= 1; s >>= 1) for (int l = 0; l < s; ++l) acc[l] += acc[l + s];
    return acc[0];
}

/* GetInfo  correlation_flow.cc:238-243 */
float ora_get_info(const float* g, long n, float response) {
    const float side_lobe_mean = (sum16(g, n) - response) / (float)(n - 1);
    float acc[16] = { 0 };
    long i = 0;
    for (; i + 16 <= n; i += 16) for (int l = 0; l < 16; ++l) { float d = g[i + l] - side_lobe_mean; acc[l] += d * d; }
    for (int l = 0; i < n; ++i, ++l) { float d = g[i] - side_lobe_mean; acc[l] += d * d; }
    for (int s = 8; s >= 1; s >>= 1) for (int l = 0; l < s; ++l) acc[l] += acc[l + s];
    const float std_ = sqrtf(acc[0] / (float)n);
    return (response - side_lobe_mean) / (std_ + 1e-7f);      /* 1e-7 is a double literal; float(std)+1e-7 then float division */
}

/* Array::pow(int) [recalled, RECALLED.md row 16 -- an OPEN question]: mode 0 (default) = std::pow(float, int) -> double pow,
 * rounded back to float; mode 1 = the exponent promoted to the array's scalar first (Eigen 3.3's promote_scalar_arg) ->
 * powf(x, (float)p).  With this image's glibc the two differ by one ulp on 6.6e-4 of the samples (tests/test_toolchain_pins.py);
 * whoever runs oracle/pin/ against a build of the reference flips the switch if the vectors say so. */
static int g_pow_mode = 0;
void ora_set_pow_mode(int mode) { g_pow_mode = mode ? 1 : 0; }
int ora_get_pow_mode(void) { return g_pow_mode; }
static inline float pow_int(float x, int p) { return g_pow_mode ? powf(x, (float)p) : (float)pow((double)x, (double)p); }

/* shared tail of the four kernel functions: kernel = kernel / kernel.abs().maxCoeff(); return FFT(kernel) */
static void normalise_and_fft(ora_ctx* ctx, float* kernel, int rows, int cols, ora_cf32* out) {
    const long n = (long)rows * cols;
    float mx = fabsf(kernel[0]);
    for (long i = 1; i < n; ++i) { float a = fabsf(kernel[i]); if (a > mx) mx = a; }
    for (long i = 0; i < n; ++i) kernel[i] = kernel[i] / mx;
    ora_fft(ctx, kernel, rows, cols, out);
}

/* polynomial_kernel  correlation_flow.cc:208-226 (xf == zf for the one-argument overload) */
static void polynomial_kernel(ora_ctx* ctx, const ora_cf32* xf, const ora_cf32* zf, int rows, int cols, ora_cf32* out) {
    const int hr = rows / 2 + 1; const size_t nc = (size_t)hr * cols; const long n = (long)rows * cols;
    ora_cf32* xzf = (ora_cf32*)tmp_alloc(sizeof(ora_cf32) * nc);
    for (size_t i = 0; i < nc; ++i) xzf[i] = cmul(xf[i], cconj(zf[i]));
    float* xz = (float*)tmp_alloc(sizeof(float) * (size_t)n);
    ora_ifft(ctx, xzf, hr, cols, xz);
    for (long i = 0; i < n; ++i) xz[i] = pow_int(xz[i] + ctx->cfg.offset, ctx->cfg.power);
    normalise_and_fft(ctx, xz, rows, cols, out);
    tmp_free(xz); tmp_free(xzf);
}

/* gaussian_kernel  correlation_flow.cc:181-206.  NOTE: xf.square().abs().sum() runs over the
 * STORED half spectrum only ((rows/2+1) x cols bins) -- not the full Parseval sum. */
static void gaussian_kernel(ora_ctx* ctx, const ora_cf32* xf, const ora_cf32* zf, int rows, int cols, ora_cf32* out) {
    const int hr = rows / 2 + 1; const size_t nc = (size_t)hr * cols; const long n = (long)rows * cols;
    const unsigned int N = (unsigned int)(rows * cols);
    float* tmp = (float*)tmp_alloc(sizeof(float) * nc);
    for (size_t i = 0; i < nc; ++i) { ora_cf32 s = cmul(xf[i], xf[i]); tmp[i] = hypotf(s.re, s.im); }
    const float xx = sum16(tmp, (long)nc) / (float)N;
    for (size_t i = 0; i < nc; ++i) { ora_cf32 s = cmul(zf[i], zf[i]); tmp[i] = hypotf(s.re, s.im); }
    const float zz = sum16(tmp, (long)nc) / (float)N;
    tmp_free(tmp);
    ora_cf32* xzf = (ora_cf32*)tmp_alloc(sizeof(ora_cf32) * nc);
    for (size_t i = 0; i < nc; ++i) xzf[i] = cmul(xf[i], cconj(zf[i]));
    float* xz = (float*)tmp_alloc(sizeof(float) * (size_t)n);
    ora_ifft(ctx, xzf, hr, cols, xz);
    const float coef = -1 / (ctx->cfg.sigma * ctx->cfg.sigma);
    for (long i = 0; i < n; ++i) {
        const float xxzz = (xx + zz - 2 * xz[i]) / (float)N;
        xz[i] = expf(coef * xxzz);
    }
    normalise_and_fft(ctx, xz, rows, cols, out);
    tmp_free(xz); tmp_free(xzf);
}

/* EstimateTrans  correlation_flow.cc:145-179 */
float ora_estimate_trans(ora_ctx* ctx, const ora_cf32* last_fft, const ora_cf32* cur_fft, int which,
                         double trans[2], int* prow, int* pcol, float* g_out, int* err) {
    const int height = which ? ctx->PD : ctx->H, width = which ? ctx->PC : ctx->W;
    const ora_cf32* output_fft = which ? ctx->target_rotation_fft : ctx->target_fft;
    const int hr = height / 2 + 1; const size_t nc = (size_t)hr * width; const long n = (long)height * width;
    if (err) *err = 0;
    if (ctx->cfg.kernel != 0 && ctx->cfg.kernel != 1) { if (err) *err = -1; return NAN; }   /* :167-168 throw */
    ora_cf32* Kzz = (ora_cf32*)tmp_alloc(sizeof(ora_cf32) * nc * 2);
    ora_cf32* Kxz = Kzz + nc;
    if (ctx->cfg.kernel == 0) {
        polynomial_kernel(ctx, last_fft, last_fft, height, width, Kzz);
        polynomial_kernel(ctx, cur_fft, last_fft, height, width, Kxz);
    } else {
        gaussian_kernel(ctx, last_fft, last_fft, height, width, Kzz);
        gaussian_kernel(ctx, cur_fft, last_fft, height, width, Kxz);
    }
    for (size_t i = 0; i < nc; ++i) {                   /* :171-172 H = T/(Kzz+lambda); G = H*Kxz */
        const ora_cf32 den = { Kzz[i].re + ctx->cfg.lambda, Kzz[i].im };
        const ora_cf32 num = output_fft[i];
        const float d = den.re * den.re + den.im * den.im;
        const ora_cf32 Hh = { (num.re * den.re + num.im * den.im) / d, (num.im * den.re - num.re * den.im) / d };
        Kzz[i] = cmul(Hh, Kxz[i]);
    }
    float* g = g_out ? g_out : (float*)tmp_alloc(sizeof(float) * (size_t)n);
    ora_ifft(ctx, Kzz, hr, width, g);
    /* :175 g.maxCoeff(&row,&col): Eigen visitor, column-major traversal, first strict max [recalled] */
    long best = 0; float response = g[0];
    if (ctx->win_radius < 0) {
        for (long i = 1; i < n; ++i) if (g[i] > response) { response = g[i]; best = i; }
    } else {
        /* extension: the same traversal, candidates only inside the cyclic window (rotation surface: also around
           the 180-degree mirror row of the window centre) */
        const int R = ctx->win_radius, wr = ctx->win_row[which ? 1 : 0], wc = ctx->win_col[which ? 1 : 0];
        best = -1; response = -INFINITY;
        for (long i = 0; i < n; ++i) {
            const int r = (int)(i % height), c = (int)(i / height);
            int dc = abs(c - wc); if (width - dc < dc) dc = width - dc;
            int dr = abs(r - wr); if (height - dr < dr) dr = height - dr;
            if (which) { const int m = abs(dr - height / 2); if (m < dr) dr = m; }
            if (dc <= R && dr <= R && g[i] > response) { response = g[i]; best = i; }
        }
    }
    const int row = (int)(best % height), col = (int)(best / height);
    trans[0] = -(row - height / 2);
    trans[1] = -(col - width / 2);
    if (prow) *prow = row;
    if (pcol) *pcol = col;
    const float info = ora_get_info(g, n, response);
    if (!g_out) tmp_free(g);
    tmp_free(Kzz);
    return info;
}

/* ComputePose  correlation_flow.cc:97-143 */
int ora_compute_pose(ora_ctx* ctx, const ora_cf32* last_fft_result, const float* image,
                     const ora_cf32* last_fft_polar, const ora_cf32* fft_polar,
                     int not_large_rotation, int faithful, double pose[3], double info[3], ora_pose_debug* dbg) {
    const int H = ctx->H, W = ctx->W; const size_t n = (size_t)H * W, nc = (size_t)(H / 2 + 1) * W;
    double trans[2] = { 0, 0 }, trans_orig[2], trans_veri[2], rots[2];
    int err = 0, rr = 0, rc = 0;
    ora_pose_debug d; memset(&d, 0, sizeof(d));
    float* grot = (float*)tmp_alloc(sizeof(float) * (size_t)ctx->PD * ctx->PC);
    const float info_rots = ora_estimate_trans(ctx, last_fft_polar, fft_polar, 1, rots, &rr, &rc, grot, &err);   /* :103 */
    if (err) { tmp_free(grot); return -1; }
    float info_rots_used = info_rots;
    if (ctx->force_rot_row >= 0) {          /* test hook: another (near-tied) position of the same surface */
        rr = ctx->force_rot_row; rc = ctx->force_rot_col;
        rots[0] = -(rr - ctx->PD / 2); rots[1] = -(rc - ctx->PC / 2);
        info_rots_used = ora_get_info(grot, (long)ctx->PD * ctx->PC, grot[(size_t)rc * ctx->PD + rr]);
        d.rot_forced = 1;
    }
    d.rot_row = rr; d.rot_col = rc; d.psr_rot = info_rots_used;
    d.rot_peak = grot[(size_t)rc * ctx->PD + rr];
    d.rot_mirror = grot[(size_t)rc * ctx->PD + (rr + ctx->PD / 2) % ctx->PD];
    tmp_free(grot);
    float degree = (float)(rots[0] * (2.0 / ctx->cfg.rotation_divisor) * 180);      /* :105 */
    degree = (float)ora_normalize_degree(degree);                                      /* :106 */
    float info_trans;
    float* rot = (float*)tmp_alloc(sizeof(float) * n);
    ora_cf32* frot = (ora_cf32*)tmp_alloc(sizeof(ora_cf32) * nc);
    if (not_large_rotation) {
        degree = fabsf(degree) > 90 ? degree - 180 : degree;                           /* :108 */
        ora_rotate(image, H, W, -degree, rot);                                          /* :109 */
        ora_fft(ctx, rot, H, W, frot);
        const float io = ora_estimate_trans(ctx, last_fft_result, frot, 0, trans_orig, &d.trans_row[0], &d.trans_col[0], NULL, &err);
        d.psr_trans[0] = io; d.degree_used[0] = -degree; d.n_hyp = 1; d.chosen = 0;
        info_trans = io; trans[0] = trans_orig[0]; trans[1] = trans_orig[1];
    } else {
        ora_rotate(image, H, W, -degree, rot);                                          /* :116 */
        ora_fft(ctx, rot, H, W, frot);
        const float io = ora_estimate_trans(ctx, last_fft_result, frot, 0, trans_orig, &d.trans_row[0], &d.trans_col[0], NULL, &err);
        ora_rotate(image, H, W, -degree + 180, rot);                                    /* :117 */
        ora_fft(ctx, rot, H, W, frot);
        const float iv = ora_estimate_trans(ctx, last_fft_result, frot, 0, trans_veri, &d.trans_row[1], &d.trans_col[1], NULL, &err);
        d.psr_trans[0] = io; d.psr_trans[1] = iv; d.degree_used[0] = -degree; d.degree_used[1] = -degree + 180; d.n_hyp = 2;
        if (io > iv) { info_trans = io; trans[0] = trans_orig[0]; trans[1] = trans_orig[1]; d.chosen = 0; }
        else { info_trans = iv; trans[0] = trans_veri[0]; trans[1] = trans_veri[1]; degree = degree + 180; d.chosen = 1; }
    }
    if (degree > 180) degree = degree - 360;                                            /* :134 */
    const float theta = (float)(degree / 180 * M_PI);                                   /* :135 */
    info[0] = info_trans; pose[0] = trans[1];
    info[1] = info_trans; pose[1] = trans[0];
    info[2] = info_rots_used;  pose[2] = theta;
    d.degree_final = degree;
    /* :139-140 std::cout omitted (I/O) */
    if (faithful) {                                                                     /* :141 dead `rectify` */
        float* back = (float*)tmp_alloc(sizeof(float) * n * 2);
        ora_ifft(ctx, last_fft_result, H / 2 + 1, W, back);
        ora_warp(back, H, W, (float)-pose[0], (float)-pose[1], degree, back + n);
        tmp_free(back);
    }
    if (dbg) *dbg = d;
    tmp_free(rot); tmp_free(frot);
    return 0;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

int ora_track_pairs(const ora_config* cfg, int H, int W, int n, const uint8_t* key_imgs, const uint8_t* cur_imgs,
                    int not_large_rotation, int faithful, int nthreads,
                    double* poses, double* infos, ora_pose_debug* dbgs, double* seconds_unit) {
    int rc = 0; double t_total = 0;
    if (nthreads < 1) nthreads = 1;
    /* plane-sized temporaries are malloc'ed per call (as the reference's Eigen temporaries are): keep them on the
       per-thread heaps instead of mmap/munmap, whose kernel lock serialises many threads */
    mallopt(M_MMAP_THRESHOLD, 1 << 30); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_ARENA_MAX, 512);
#ifdef _OPENMP
    omp_set_num_threads(nthreads);
#endif
    double t_wall0 = 0, t_key = 0;
    #pragma omp parallel reduction(+:t_key)
    {
        ora_ctx* ctx = ora_create(cfg, H, W);                  /* per-thread context: set-up is not part of the timed units */
        if (ctx) tmp_reserve((size_t)48 * sizeof(ora_cf32) * ((size_t)(H / 2 + 1) * W + (size_t)(ctx->PD / 2 + 1) * ctx->PC));
        const size_t npx = (size_t)H * W, nc = (size_t)(H / 2 + 1) * W;
        const size_t ncp = ctx ? (size_t)(ctx->PD / 2 + 1) * ctx->PC : 0;
        float* img = (float*)malloc(sizeof(float) * npx);
        ora_cf32* kf = (ora_cf32*)malloc(sizeof(ora_cf32) * (nc + ncp) * 2);
        ora_cf32* kp = kf + nc; ora_cf32* cf = kp + ncp; ora_cf32* cp = cf + nc;
        if (ctx) { get_plan(ctx, H / 2); get_plan(ctx, H); get_plan(ctx, W); get_plan(ctx, ctx->PD / 2); get_plan(ctx, ctx->PD); get_plan(ctx, ctx->PC); }
        #pragma omp barrier
        #pragma omp master
        t_wall0 = now_s();
        #pragma omp barrier
        #pragma omp for schedule(dynamic, 1)
        for (int i = 0; i < n; ++i) {
            if (!ctx) { rc = -2; continue; }
            const double t0 = now_s();
            ora_normalize_u8(key_imgs + (size_t)i * npx, H, W, img);
            ora_intermedium(ctx, img, kf, kp);
            t_key += now_s() - t0;
            ora_normalize_u8(cur_imgs + (size_t)i * npx, H, W, img);       /* MapBuilder::ComputeFFTResult map_builder.cc:72-75 */
            ora_intermedium(ctx, img, cf, cp);
            if (ora_compute_pose(ctx, kf, img, kp, cp, not_large_rotation, faithful,
                                 poses + 3 * i, infos + 3 * i, dbgs ? dbgs + i : NULL)) rc = -1;
        }
        #pragma omp master
        t_total = now_s() - t_wall0;
        free(img); free(kf); ora_destroy(ctx); tmp_release();
    }
    /* wall time of the timed units = total wall minus the (thread-averaged) key preparation */
    if (seconds_unit) *seconds_unit = t_total - t_key / nthreads;
    return rc;
}

/* ======================================================================================================
 * Camera undistortion (camera.cc:45-47, 92-93)
 * ====================================================================================================== */

/* cvUndistortPoints (C API default criteria: 5 fixed-point iterations), no R / P: pixel -> normalised coords */
static void undistort_point(double u, double v, const double K[4], const double D[5], double* xo, double* yo) {
    const double fx = K[0], cx = K[1], fy = K[2], cy = K[3];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
        if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
        const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
        const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    *xo = x; *yo = y;
}

/* cvGetOptimalNewCameraMatrix(alpha = 0, newImgSize = imgSize, centerPrincipalPoint = false): the inscribed
 * rectangle of a 9x9 grid of undistorted points (icvGetRectangles; points are stored as float) mapped to the viewport */
void ora_optimal_new_camera_matrix(const double K[4], const double D[5], int width, int height, double newK[4]) {
    const int N = 9;
    float iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX;
    for (int y = 0; y < N; ++y)
        for (int x = 0; x < N; ++x) {
            const float px = (float)x * width / (N - 1), py = (float)y * height / (N - 1);
            double ux, uy;
            undistort_point((double)px, (double)py, K, D, &ux, &uy);
            const float qx = (float)ux, qy = (float)uy;
            if (x == 0)     iX0 = iX0 > qx ? iX0 : qx;
            if (x == N - 1) iX1 = iX1 < qx ? iX1 : qx;
            if (y == 0)     iY0 = iY0 > qy ? iY0 : qy;
            if (y == N - 1) iY1 = iY1 < qy ? iY1 : qy;
        }
    const float inner_x = iX0, inner_y = iY0, inner_w = iX1 - iX0, inner_h = iY1 - iY0;   /* cv::Rect_<float> */
    const double fx0 = (width - 1) / inner_w, fy0 = (height - 1) / inner_h;
    newK[0] = fx0; newK[1] = -fx0 * inner_x; newK[2] = fy0; newK[3] = -fy0 * inner_y;
}

static inline int sat_int_d(double v) {      /* saturate_cast<int>(double) = cvRound with saturation */
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return cv_round_d(v);
}

/* cv::initUndistortRectifyMap(K, D, R = I, newK, size, CV_16SC2): per destination pixel the distorted source
 * position in 1/32 px fixed point */
void ora_undistort_maps(const double K[4], const double D[5], const double newK[4], int width, int height,
                        int16_t* map1, uint16_t* map2) {
    const double fx = K[0], u0 = K[1], fy = K[2], v0 = K[3];
    const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
    /* iR = (newK * I)^-1 for the upper-triangular [[fx',0,cx'],[0,fy',cy'],[0,0,1]] */
    const double ir[9] = { 1. / newK[0], 0, -newK[1] / newK[0],  0, 1. / newK[2], -newK[3] / newK[2],  0, 0, 1 };
    for (int i = 0; i < height; ++i) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < width; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double w = 1. / _w, x = _x * w, y = _y * w;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((0 * r2 + 0) * r2 + 0) * r2);
            const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
            const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
            const double u = fx * xd + u0, v = fy * yd + v0;
            const int iu = sat_int_d(u * INTER_TAB_SIZE), iv = sat_int_d(v * INTER_TAB_SIZE);
            map1[((size_t)i * width + j) * 2 + 0] = (int16_t)(iu >> INTER_BITS);
            map1[((size_t)i * width + j) * 2 + 1] = (int16_t)(iv >> INTER_BITS);
            map2[(size_t)i * width + j] = (uint16_t)((iv & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (iu & (INTER_TAB_SIZE - 1)));
        }
    }
}

/* cv::remap(8UC1, CV_16SC2 + CV_16UC1 maps, INTER_LINEAR, BORDER_CONSTANT 0): fixed-point bilinear with the
 * 15-bit integer weight table (INTER_REMAP_COEF_BITS); for 1/32 px fractions the weights are exact multiples of 32
 * and sum to 32768, so they are computed directly.  Out-of-image taps contribute the border value 0. */
void ora_remap_u8(const uint8_t* src, int width, int height, const int16_t* map1, const uint16_t* map2, uint8_t* dst) {
    for (int i = 0; i < height; ++i)
        for (int j = 0; j < width; ++j) {
            const int sx = map1[((size_t)i * width + j) * 2], sy = map1[((size_t)i * width + j) * 2 + 1];
            const int m = map2[(size_t)i * width + j] & (INTER_TAB_SIZE * INTER_TAB_SIZE - 1);
            const int fx = m & (INTER_TAB_SIZE - 1), fy = m >> INTER_BITS;
            const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            int acc = 0;
            if (sy >= 0 && sy < height) {
                if (sx >= 0 && sx < width)         acc += w00 * src[(size_t)sy * width + sx];
                if (sx + 1 >= 0 && sx + 1 < width) acc += w01 * src[(size_t)sy * width + sx + 1];
            }
            if (sy + 1 >= 0 && sy + 1 < height) {
                if (sx >= 0 && sx < width)         acc += w10 * src[(size_t)(sy + 1) * width + sx];
                if (sx + 1 >= 0 && sx + 1 < width) acc += w11 * src[(size_t)(sy + 1) * width + sx + 1];
            }
            dst[(size_t)i * width + j] = (uint8_t)((acc + (1 << 14)) >> 15);   /* FixedPtCast<int, uchar, 15> */
        }
}

/* ======================================================================================================
 * Coarse-to-fine extension (BASELINE config 3; NO reference counterpart -- defined in SURVEY 8(d))
 * ====================================================================================================== */
void ora_force_rotation(ora_ctx* ctx, int row, int col) { ctx->force_rot_row = row; ctx->force_rot_col = col; }

void ora_set_window(ora_ctx* ctx, int rot_row, int rot_col, int trans_row, int trans_col, int radius) {
    ctx->win_row[1] = rot_row; ctx->win_col[1] = rot_col; ctx->win_row[0] = trans_row; ctx->win_col[0] = trans_col;
    ctx->win_radius = radius;                /* radius < 0: plain global arg-max again */
}

/* 2 x 2 box filter, rounded: u8 row-major (height x width) -> (height/2 x width/2) */
void ora_downsample_u8(const uint8_t* src, int width, int height, uint8_t* dst) {
    const int wo = width / 2, ho = height / 2;
    for (int r = 0; r < ho; ++r)
        for (int c = 0; c < wo; ++c) {
            const uint8_t* p = src + (size_t)(2 * r) * width + 2 * c;
            dst[(size_t)r * wo + c] = (uint8_t)((p[0] + p[1] + p[width] + p[width + 1] + 2) >> 2);
        }
}
