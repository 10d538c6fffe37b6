// kcc_api.hip -- C ABI of libnislam_kcc_hip.so (see include/nislam_kcc.h).
// Host side: context, device-resident keyframe store, batched stage scheduling on one HIP stream.
// Mirrors CorrelationFlow (reference include/correlation_flow.h:8-33, src/correlation_flow.cc:37-143).
#include "../../include/nislam_kcc.h"
#include "kcc_kernels.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace kcc;

namespace {

thread_local std::string g_create_error;

struct Family {                 // one plane geometry with its tables
    PlaneGeom g{};
    Tables t{};
    size_t real_elems = 0;      // rows*cols
    size_t spec_elems = 0;      // hr*cols
    float2* d_tw[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
};

}  // namespace

struct nik_ctx {
    nik_config cfg{};
    int H = 0, W = 0, PD = 0, PC = 0;
    int max_batch = 0, max_frames = 0, device = 0;
    int max_items = 0;          // 2*max_batch (two hypotheses per pair in large-rotation mode)
    hipStream_t stream = nullptr;
    hipEvent_t idx_event = nullptr;
    std::string err;

    Family img, pol;
    // keyframe store (reference Frame: _frame, _fft_result, _fft_polar)
    float* arena_img = nullptr; float2* arena_F = nullptr; float2* arena_P = nullptr;
    std::vector<uint8_t> slot_ready;     // bit0: image, bit1: spectra
    // work buffers
    float2* tmpA = nullptr;              // [max_items][max spec]
    float2* kbuf = nullptr;              // [max_items][2][max spec]  (zz, xz planes)
    float2* gbuf = nullptr;              // [max_items][max spec]
    float*  splane = nullptr;            // [max_batch][(W+1)*(H+2)] shifted zero-bordered planes (polar source)
    size_t  s_elems = 0;
    size_t  spec_max = 0;
    Partial* partials = nullptr; int partial_stride = 0;
    unsigned* maxbuf = nullptr;          // [max_items][2]
    float* energy = nullptr;             // [max_items][2]
    SurfaceResult* rot_res = nullptr;    // [max_batch]
    SurfaceResult* trans_res = nullptr;  // [max_items]
    int* d_idx = nullptr;                // device int scratch: 6 arrays of max_items
    int* h_idx = nullptr;                // pinned mirror
    SurfaceResult* h_rot = nullptr; SurfaceResult* h_trans = nullptr;   // pinned
    uint8_t* d_u8 = nullptr;             // staging for host u8 input (one image)
    float* d_scratch = nullptr;          // debug / import-export staging (max(real, 2*spec) floats)
    uint32_t* polar_tab = nullptr;
    int* rot_tab = nullptr;              // [3][PD][2W+2H] fixed-point warpAffine terms per candidate angle
    std::vector<float> rot_deg;          // [3][PD] degree after normalise/fold (variant 0) or hypothesis angles
    // per-stage HIP-event profiler (nik_profile_enable / nik_profile_read)
    struct StageStat { std::string name; double ms = 0; long launches = 0; double bytes = 0; };
    struct StageRec { int stage; hipEvent_t a, b; };
    bool prof_on = false;
    std::vector<StageStat> prof_stats;
    std::vector<StageRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    // pending asynchronous batch
    struct Pending { bool active = false; int n = 0; int n_hyp = 1; nik_pose_result* res = nullptr; } pending;
};

namespace {

int fail(nik_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail(c, NIK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

std::vector<float2> twiddles(int n, int count) {
    std::vector<float2> t(count);
    for (int i = 0; i < count; ++i) {
        const double a = -2.0 * M_PI * (double)i / (double)n;
        t[i] = make_float2((float)cos(a), (float)sin(a));
    }
    return t;
}

// pass-twiddle table of one plan and direction ("Twiddle table layout" in kcc_fft2.h); evaluated in double
std::vector<float2> plan_table(const PlanDesc& d, bool inv) {
    const int N = d.n;
    int r[3] = { d.r[0], d.r[1], d.r[2] };
    if (inv) { if (d.np == 3) std::swap(r[0], r[2]); else std::swap(r[0], r[1]); }
    const int RF = r[0], RM = d.np == 3 ? r[1] : 1, RL = d.np == 3 ? r[2] : r[1];
    const double sgn = inv ? 2.0 * M_PI : -2.0 * M_PI;
    auto w = [&](long num, long den) { const double a = sgn * (double)(num % den) / (double)den; return make_float2((float)cos(a), (float)sin(a)); };
    std::vector<float2> t;
    if (d.np == 2) {
        t.resize((size_t)RL * RF);
        for (int q = 0; q < RL; ++q) for (int k = 0; k < RF; ++k) t[(size_t)q * RF + k] = w((long)q * k, N);
    } else {
        const int NS = RF * RM;
        t.resize((size_t)RM * RF + (size_t)RL * NS);
        for (int q = 0; q < RM; ++q) for (int k = 0; k < RF; ++k) t[(size_t)q * RF + k] = w((long)q * k, NS);
        for (int q = 0; q < RL; ++q) for (int k = 0; k < NS; ++k) t[(size_t)RM * RF + (size_t)q * NS + k] = w((long)q * k, N);
    }
    return t;
}

int upload_table(nik_ctx* c, const std::vector<float2>& h, float2** d) {
    HIP_TRY(c, hipMalloc(d, sizeof(float2) * h.size()));
    HIP_TRY(c, hipMemcpy(*d, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice));
    return NIK_OK;
}

int family_init(nik_ctx* c, Family& f, int rows, int cols) {
    f.g.rows = rows; f.g.cols = cols; f.g.hr = rows / 2 + 1;
    f.real_elems = (size_t)rows * cols; f.spec_elems = (size_t)f.g.hr * cols;
    const int h = rows / 2;
    const PlanDesc ph = plan_desc(h), pc = plan_desc(cols);
    int rc;
    if ((rc = upload_table(c, plan_table(ph, false), &f.d_tw[0])) || (rc = upload_table(c, plan_table(ph, true), &f.d_tw[1])) ||
        (rc = upload_table(c, twiddles(rows, h), &f.d_tw[2])) ||
        (rc = upload_table(c, plan_table(pc, false), &f.d_tw[3])) || (rc = upload_table(c, plan_table(pc, true), &f.d_tw[4]))) return rc;
    f.t.half_f = f.d_tw[0]; f.t.half_i = f.d_tw[1]; f.t.tw_full = f.d_tw[2]; f.t.cols_f = f.d_tw[3]; f.t.cols_i = f.d_tw[4];
    return NIK_OK;
}

inline int cv_round_f(float v) { return (int)lrintf(v); }

// cv::warpPolar map (reference correlation_flow.cc:231-234) quantised as cv::remap does (1/32 px), stored
// [PC][PD] so a polar line (fixed radius, all angles) is contiguous.  Entry: (sx*(H+2)+sy) | fx<<22 | fy<<27.
int build_polar_table(nik_ctx* c) {
    const int PD = c->PD, PC = c->PC, H = c->H, W = c->W;
    std::vector<uint32_t> tab((size_t)PD * PC);
    const float cx = (float)W / 2, cy = (float)H / 2;
    const double maxRadius = (double)std::min(H / 2, W / 2);
    const double Kangle = (2.0 * 3.1415926535897932384626433832795) / PD;
    const double Kmag = maxRadius / PC;
    std::vector<float> rhos(PC);
    for (int rho = 0; rho < PC; ++rho) rhos[rho] = (float)(rho * Kmag);
    for (int phi = 0; phi < PD; ++phi) {
        const double KKy = Kangle * phi;
        const double cp = cos(KKy), sp = sin(KKy);
        for (int rho = 0; rho < PC; ++rho) {
            const float mx = (float)(rhos[rho] * cp + cx);
            const float my = (float)(rhos[rho] * sp + cy);
            const int qx = cv_round_f(mx * 32), qy = cv_round_f(my * 32);
            const int sx = qx >> 5, sy = qy >> 5;
            // all four taps must fall inside the zero-bordered plane S[W+1][H+2] (taps beyond the image read 0,
            // exactly cv::remap's BORDER_CONSTANT path for a source that never leaves the image by more than 1 px)
            if (sx < 0 || sy < 0 || sx + 1 > W || sy + 1 > H)
                return fail(c, NIK_ERR_UNSUPPORTED_SIZE, "polar map leaves the image by more than one pixel");
            const uint32_t off = (uint32_t)sx * (uint32_t)(H + 2) + (uint32_t)sy;
            if (off >= (1u << 22)) return fail(c, NIK_ERR_UNSUPPORTED_SIZE, "image too large for the packed polar table");
            tab[(size_t)rho * PD + phi] = off | ((uint32_t)(qx & 31) << 22) | ((uint32_t)(qy & 31) << 27);
        }
    }
    HIP_TRY(c, hipMalloc(&c->polar_tab, sizeof(uint32_t) * tab.size()));
    HIP_TRY(c, hipMemcpy(c->polar_tab, tab.data(), sizeof(uint32_t) * tab.size(), hipMemcpyHostToDevice));
    return NIK_OK;
}

double normalize_degree(double a) { return a - 360 * floor((a + 180) / 360); }     // utils.cc:173-175

// Fixed-point terms of cv::warpAffine (WarpAffineInvoker, INTER_LINEAR) for RotateArray(image, degree_arg)
// (utils.cc:154-161): the inverse of getRotationMatrix2D(center, angle, 1) in double, then
//   adelta[c] = rint(M0*c*1024), bdelta[c] = rint(M3*c*1024), X0[r] = rint((M1*r+M2)*1024)+16, Y0[r] likewise.
// Layout: [adelta W | bdelta W | X0 H | Y0 H].
void rotation_terms(int H, int W, float degree_arg, int* out) {
    const float cx = (float)(W / 2.), cy = (float)(H / 2.);
    double angle = (double)degree_arg;
    angle *= 3.1415926535897932384626433832795 / 180;
    const double alpha = cos(angle) * 1.0, beta = sin(angle) * 1.0;
    double M[6] = { alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy };
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    const int round_delta = 1024 / 32 / 2;
    for (int c = 0; c < W; ++c) { out[c] = (int)lrint(M[0] * c * 1024); out[W + c] = (int)lrint(M[3] * c * 1024); }
    for (int r = 0; r < H; ++r) {
        out[2 * W + r] = (int)lrint((M[1] * r + M[2]) * 1024) + round_delta;
        out[2 * W + H + r] = (int)lrint((M[4] * r + M[5]) * 1024) + round_delta;
    }
}

// For every possible rotation arg-max row: the angles ComputePose feeds to RotateArray
// (correlation_flow.cc:105-117).  variant 0: not_large_rotation; 1: `orig`; 2: `veri` (+180).
int build_rot_table(nik_ctx* c) {
    const int PD = c->PD;
    const size_t per = (size_t)2 * c->W + 2 * c->H;
    std::vector<int> tab((size_t)3 * PD * per);
    c->rot_deg.assign((size_t)3 * PD, 0.f);
    for (int row = 0; row < PD; ++row) {
        const double rots0 = -(row - PD / 2);
        float degree = (float)(rots0 * (2.0 / c->cfg.rotation_divisor) * 180);       // :105
        degree = (float)normalize_degree(degree);                                       // :106
        const float d0 = std::abs(degree) > 90 ? degree - 180 : degree;                 // :108
        c->rot_deg[0 * PD + row] = d0;       rotation_terms(c->H, c->W, -d0, &tab[(0 * PD + row) * per]);
        c->rot_deg[1 * PD + row] = degree;   rotation_terms(c->H, c->W, -degree, &tab[((size_t)1 * PD + row) * per]);
        c->rot_deg[2 * PD + row] = degree;   rotation_terms(c->H, c->W, -degree + 180, &tab[((size_t)2 * PD + row) * per]);
    }
    HIP_TRY(c, hipMalloc(&c->rot_tab, sizeof(int) * tab.size()));
    HIP_TRY(c, hipMemcpy(c->rot_tab, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice));
    return NIK_OK;
}

// index scratch layout (each max_items ints)
enum { IX_KEY = 0, IX_CUR = 1, IX_DST = 2, IX_PAIR = 3, IX_VARIANT = 4, IX_TIMG = 5, IX_TKEY = 6, IX_ROTIDX = 7, IX_COUNT = 8 };
inline int* didx(nik_ctx* c, int which) { return c->d_idx + (size_t)which * c->max_items; }
inline int* hidx(nik_ctx* c, int which) { return c->h_idx + (size_t)which * c->max_items; }

// The pinned staging arrays are reused by every call: wait until the previous uploads have been consumed
// (a tiny H2D copy each -- this never waits for the kernels queued behind them).
int begin_idx(nik_ctx* c) {
    HIP_TRY(c, hipEventSynchronize(c->idx_event));
    return NIK_OK;
}
int upload_idx(nik_ctx* c, int which, int n) {
    HIP_TRY(c, hipMemcpyAsync(didx(c, which), hidx(c, which), sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(c->idx_event, c->stream));
    return NIK_OK;
}

int check_slot(nik_ctx* c, nik_frame f, bool need_ready) {
    if (f < 0 || f >= c->max_frames) return fail(c, NIK_ERR_CAPACITY, "frame slot %d out of range [0,%d)", f, c->max_frames);
    if (need_ready && c->slot_ready[f] != 3) return fail(c, NIK_ERR_NOT_READY, "frame slot %d holds no image/spectra", f);
    return NIK_OK;
}

KernelFn kernel_fn(const nik_ctx* c) {
    KernelFn fn; fn.type = c->cfg.kernel; fn.offset = c->cfg.offset; fn.power = c->cfg.power; fn.sigma = c->cfg.sigma;
    fn.lambda = c->cfg.lambda;
    return fn;
}

// ---- stage profiler: brackets one kernel launch with HIP events on the launch stream -----------------
struct Stage {
    nik_ctx* c; int rec = -1;
    Stage(nik_ctx* c_, const char* name, double bytes) : c(c_) {
        if (!c->prof_on) return;
        int id = -1;
        for (size_t i = 0; i < c->prof_stats.size(); ++i) if (c->prof_stats[i].name == name) { id = (int)i; break; }
        if (id < 0) { c->prof_stats.push_back({}); id = (int)c->prof_stats.size() - 1; c->prof_stats[id].name = name; }
        c->prof_stats[id].launches += 1; c->prof_stats[id].bytes += bytes;
        nik_ctx::StageRec r; r.stage = id;
        for (hipEvent_t* e : { &r.a, &r.b }) {
            if (!c->prof_pool.empty()) { *e = c->prof_pool.back(); c->prof_pool.pop_back(); }
            else if (hipEventCreate(e) != hipSuccess) return;
        }
        (void)hipEventRecord(r.a, c->stream);
        c->prof_recs.push_back(r); rec = (int)c->prof_recs.size() - 1;
    }
    ~Stage() { if (rec >= 0) (void)hipEventRecord(c->prof_recs[rec].b, c->stream); }
};
std::string kname(const char* base, int len, const char* mode) {
    char b[64]; snprintf(b, sizeof(b), "%s<%d,%s>", base, len, mode); return b;
}
inline double Rb(const Family& f) { return 4.0 * (double)f.real_elems; }     // real plane bytes
inline double Cb(const Family& f) { return 8.0 * (double)f.spec_elems; }     // half-spectrum plane bytes

// ComputeIntermedium (correlation_flow.cc:89-95) for n images already stored (f32, column-major) in the
// arena slots listed in d_idx[IX_DST].
void enqueue_intermedium(nik_ctx* c, int n) {
    hipStream_t s = c->stream;
    const int* dst = didx(c, IX_DST);
    const Family& I = c->img; const Family& P = c->pol;
    { Stage st(c, kname("kA_fwd", c->H / 2, "plane").c_str(), n * (Rb(I) + Cb(I)));
      launch_A_fwd_plane(s, n, c->img.g, c->img.t, c->arena_img, c->img.real_elems, dst, c->tmpA, c->spec_max); }
    { Stage st(c, kname("kB", c->W, "fwd_abs_inv").c_str(), n * 3 * Cb(I));
      launch_B_fwd_abs_inv(s, n, c->img.g, c->img.t, c->tmpA, c->spec_max, c->arena_F, c->img.spec_elems, dst,
                           c->gbuf, c->spec_max); }
    { Stage st(c, kname("kA_inv", c->H / 2, "shifted").c_str(), n * (Cb(I) + Rb(I)));
      launch_A_inv_shifted(s, n, c->img.g, c->img.t, c->gbuf, c->spec_max, c->splane, c->s_elems); }
    launch_fix_zero(s, n, c->splane, c->s_elems, c->H, c->W);
    { Stage st(c, kname("kA_fwd", c->PD / 2, "polar").c_str(), n * (Rb(I) + Cb(P)) + 4.0 * c->PD * c->PC);
      launch_A_fwd_polar(s, n, c->pol.g, c->pol.t, c->splane, c->s_elems, c->H, c->W, c->polar_tab,
                         c->tmpA, c->spec_max); }
    { Stage st(c, kname("kB", c->PC, "fwd").c_str(), n * 2 * Cb(P));
      launch_B_fwd(s, n, c->pol.g, c->pol.t, c->tmpA, c->spec_max, c->arena_P, c->pol.spec_elems, dst); }
}

// EstimateTrans (correlation_flow.cc:145-179) for n items.  X spectra: x_fwd ? forward of tmpA lines : arena.
void enqueue_estimate(nik_ctx* c, int n, Family& f, bool x_fwd, const float2* xsrc, size_t x_stride, const int* x_idx,
                      const float2* zsrc, size_t z_stride, const int* z_idx, SurfaceResult* out) {
    hipStream_t s = c->stream;
    const size_t item_stride = 2 * c->spec_max, plane_stride = c->spec_max;
    (void)hipMemsetAsync(c->maxbuf, 0, sizeof(unsigned) * 2 * n, s);
    if (c->cfg.kernel == 1 && !x_fwd)
        launch_energy(s, n, f.g, xsrc, x_stride, x_idx, zsrc, z_stride, z_idx, c->energy);
    { Stage st(c, kname("kB", f.g.cols, x_fwd ? "fwd_mul_inv" : "mul_inv").c_str(), n * 4 * Cb(f));
      launch_B_mul_inv(s, n, f.g, f.t, x_fwd, xsrc, x_stride, x_idx, zsrc, z_stride, z_idx, c->kbuf, item_stride, plane_stride); }
    { Stage st(c, kname("kA_inv", f.g.rows / 2, "kernel_fwd").c_str(), n * 4 * Cb(f));
      launch_A_inv_kernel_fwd(s, n, f.g, f.t, c->kbuf, item_stride, plane_stride, kernel_fn(c), c->maxbuf, c->energy); }
    { Stage st(c, kname("kB", f.g.cols, "solve_inv").c_str(), n * 3 * Cb(f));
      launch_B_solve_inv(s, n, f.g, f.t, c->kbuf, item_stride, plane_stride, c->maxbuf, c->cfg.lambda, c->gbuf, c->spec_max); }
    const int nb = argmax_blocks(f.g);
    { Stage st(c, kname("kA_inv", f.g.rows / 2, "argmax").c_str(), n * Cb(f));
      launch_A_inv_argmax(s, n, f.g, f.t, c->gbuf, c->spec_max, c->partials, c->partial_stride); }
    launch_finalize(s, n, c->partials, c->partial_stride, nb, out);
}

// ComputePose (correlation_flow.cc:97-143) for n pairs; key/cur slots in d_idx[IX_KEY]/[IX_CUR].
// Leaves raw surface results in h_rot / h_trans (valid after the stream is synchronised).
int enqueue_pose(nik_ctx* c, int n, int not_large_rotation) {
    hipStream_t s = c->stream;
    const int n_hyp = not_large_rotation ? 1 : 2, nt = n * n_hyp;
    // rotation stage: z = key polar spectrum, x = current polar spectrum
    enqueue_estimate(c, n, c->pol, false, c->arena_P, c->pol.spec_elems, didx(c, IX_CUR),
                     c->arena_P, c->pol.spec_elems, didx(c, IX_KEY), c->rot_res);
    // translation items (one per pair and hypothesis)
    for (int t = 0; t < nt; ++t) {
        const int p = t / n_hyp, hyp = t % n_hyp;
        hidx(c, IX_PAIR)[t] = p;
        hidx(c, IX_VARIANT)[t] = not_large_rotation ? 0 : 1 + hyp;
        hidx(c, IX_TIMG)[t] = hidx(c, IX_CUR)[p];
        hidx(c, IX_TKEY)[t] = hidx(c, IX_KEY)[p];
    }
    int rc;
    if ((rc = upload_idx(c, IX_PAIR, nt)) || (rc = upload_idx(c, IX_VARIANT, nt)) ||
        (rc = upload_idx(c, IX_TIMG, nt)) || (rc = upload_idx(c, IX_TKEY, nt))) return rc;
    launch_rot_index(s, nt, c->rot_res, didx(c, IX_PAIR), didx(c, IX_VARIANT), c->PD, didx(c, IX_ROTIDX));
    // FFT(RotateArray(image, -degree))  (:109 / :116-117): A pass with the rotation gather fused into its load
    { Stage st(c, kname("kA_fwd", c->H / 2, "rot").c_str(), nt * (Rb(c->img) + Cb(c->img)));
      launch_A_fwd_rot(s, nt, c->img.g, c->img.t, c->arena_img, c->img.real_elems, didx(c, IX_TIMG), c->rot_tab,
                       didx(c, IX_ROTIDX), c->tmpA, c->spec_max); }
    if (c->cfg.kernel == 1) {
        // gaussian needs sum|X|^2 of the rotated image's spectrum: materialise X (B forward, in place) first
        launch_B_fwd(s, nt, c->img.g, c->img.t, c->tmpA, c->spec_max, c->tmpA, c->spec_max, nullptr);
        enqueue_estimate(c, nt, c->img, false, c->tmpA, c->spec_max, nullptr, c->arena_F, c->img.spec_elems,
                         didx(c, IX_TKEY), c->trans_res);
    } else {
        enqueue_estimate(c, nt, c->img, true, c->tmpA, c->spec_max, nullptr, c->arena_F, c->img.spec_elems,
                         didx(c, IX_TKEY), c->trans_res);
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_rot, c->rot_res, sizeof(SurfaceResult) * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(c->h_trans, c->trans_res, sizeof(SurfaceResult) * nt, hipMemcpyDeviceToHost, s));
    return NIK_OK;
}

// GetInfo (correlation_flow.cc:238-243) from single-pass moments
float psr_from(const SurfaceResult& r, long n) {
    const double m = (r.sum - (double)r.peak) / (double)(n - 1);
    double var = (r.sumsq - 2.0 * m * r.sum + (double)n * m * m) / (double)n;
    if (var < 0) var = 0;
    return (float)(((double)r.peak - m) / ((double)(float)sqrt(var) + 1e-7));
}

// host tail of ComputePose (:105-138) from the raw arg-max results
void finalize_pose(nik_ctx* c, int i, int n_hyp, nik_pose_result* out) {
    const int PD = c->PD, H = c->H, W = c->W;
    nik_pose_result r; memset(&r, 0, sizeof(r));
    const SurfaceResult& rr = c->h_rot[i];
    r.rot_row = rr.idx % PD; r.rot_col = rr.idx / PD;
    r.psr_rot = psr_from(rr, (long)PD * c->PC);
    r.n_hyp = n_hyp;
    float degree; float info_trans; double trans0, trans1;
    auto tr = [&](int hyp, double& t0, double& t1) {
        const SurfaceResult& s = c->h_trans[i * n_hyp + hyp];
        r.trans_row[hyp] = s.idx % H; r.trans_col[hyp] = s.idx / H;
        r.psr_trans[hyp] = psr_from(s, (long)H * W);
        t0 = -(r.trans_row[hyp] - H / 2); t1 = -(r.trans_col[hyp] - W / 2);
    };
    if (n_hyp == 1) {
        degree = c->rot_deg[0 * PD + r.rot_row];
        tr(0, trans0, trans1); info_trans = r.psr_trans[0]; r.chosen = 0;
    } else {
        degree = c->rot_deg[1 * PD + r.rot_row];
        double a0, a1, b0, b1; tr(0, a0, a1); tr(1, b0, b1);
        if (r.psr_trans[0] > r.psr_trans[1]) { info_trans = r.psr_trans[0]; trans0 = a0; trans1 = a1; r.chosen = 0; }
        else { info_trans = r.psr_trans[1]; trans0 = b0; trans1 = b1; degree = degree + 180; r.chosen = 1; }
    }
    if (degree > 180) degree = degree - 360;                    // :134
    const float theta = (float)(degree / 180 * M_PI);           // :135
    r.info[0] = info_trans; r.pose[0] = trans1;
    r.info[1] = info_trans; r.pose[1] = trans0;
    r.info[2] = r.psr_rot;  r.pose[2] = theta;
    r.degree_final = degree;
    *out = r;
}

int drain_pending(nik_ctx* c) {
    if (!c->pending.active) return NIK_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->pending.res)
        for (int i = 0; i < c->pending.n; ++i) finalize_pose(c, i, c->pending.n_hyp, c->pending.res + i);
    c->pending.active = false;
    return NIK_OK;
}

int check_kernel(nik_ctx* c) {
    if (c->cfg.kernel != 0 && c->cfg.kernel != 1) return fail(c, NIK_ERR_INVALID_KERNEL, "Received invalid kernel type");
    return NIK_OK;
}

}  // namespace

extern "C" {

int nik_create(const nik_config* cfg, int image_height, int image_width, int max_batch, int max_frames, int device,
               nik_ctx** out) {
    if (!cfg || !out) return fail(nullptr, NIK_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (image_height <= 0 || image_width <= 0 || max_batch <= 0 || max_frames <= 0)
        return fail(nullptr, NIK_ERR_INVALID_ARG, "non-positive size");
    const int H = image_height, W = image_width, PD = cfg->rotation_divisor, PC = cfg->rotation_channel;
    if ((H & 1) || (W & 1) || (PD & 1) || (PC & 1) || PD <= 0 || PC <= 0)
        return fail(nullptr, NIK_ERR_UNSUPPORTED_SIZE, "height, width, rotation_divisor and rotation_channel must be even");
    if (!fft_half_supported(H / 2) || !fft_half_supported(PD / 2) || !fft_line_supported(W) || !fft_line_supported(PC))
        return fail(nullptr, NIK_ERR_UNSUPPORTED_SIZE, "FFT length not instantiated for %dx%d / polar %dx%d", H, W, PD, PC);
    if (W % 16 || PC % 16) return fail(nullptr, NIK_ERR_UNSUPPORTED_SIZE, "width and rotation_channel must be multiples of 16");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, NIK_ERR_HIP, "no HIP device available (the HIP path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, NIK_ERR_INVALID_ARG, "device %d out of range", device);
    nik_ctx* c = new nik_ctx();
    c->cfg = *cfg; c->cfg.height = H; c->cfg.width = W;      // correlation_flow.cc:40-41
    c->H = H; c->W = W; c->PD = PD; c->PC = PC; c->max_batch = max_batch; c->max_frames = max_frames; c->device = device;
    c->max_items = 2 * max_batch;
    auto bail = [&](int rc) { g_create_error = c->err; nik_destroy(c); return rc; };
#define TRY_C(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fail(c, NIK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); return bail(NIK_ERR_HIP); } } while (0)
    TRY_C(hipSetDevice(device));
    TRY_C(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    TRY_C(hipEventCreateWithFlags(&c->idx_event, hipEventDisableTiming));
    int rc;
    if ((rc = family_init(c, c->img, H, W)) || (rc = family_init(c, c->pol, PD, PC))) return bail(rc);
    c->spec_max = std::max(c->img.spec_elems, c->pol.spec_elems);
    TRY_C(hipMalloc(&c->arena_img, sizeof(float) * c->img.real_elems * max_frames));
    TRY_C(hipMalloc(&c->arena_F, sizeof(float2) * c->img.spec_elems * max_frames));
    TRY_C(hipMalloc(&c->arena_P, sizeof(float2) * c->pol.spec_elems * max_frames));
    c->slot_ready.assign(max_frames, 0);
    TRY_C(hipMalloc(&c->tmpA, sizeof(float2) * c->spec_max * c->max_items));
    TRY_C(hipMalloc(&c->kbuf, sizeof(float2) * c->spec_max * 2 * c->max_items));
    TRY_C(hipMalloc(&c->gbuf, sizeof(float2) * c->spec_max * c->max_items));
    c->s_elems = (size_t)(W + 1) * (H + 2);
    TRY_C(hipMalloc(&c->splane, sizeof(float) * c->s_elems * max_batch));
    TRY_C(hipMemset(c->splane, 0, sizeof(float) * c->s_elems * max_batch));      // zero borders are never overwritten
    c->partial_stride = std::max(argmax_blocks(c->img.g), argmax_blocks(c->pol.g));
    TRY_C(hipMalloc(&c->partials, sizeof(Partial) * c->partial_stride * c->max_items));
    TRY_C(hipMalloc(&c->maxbuf, sizeof(unsigned) * 2 * c->max_items));
    TRY_C(hipMalloc(&c->energy, sizeof(float) * 2 * c->max_items));
    TRY_C(hipMemset(c->energy, 0, sizeof(float) * 2 * c->max_items));
    TRY_C(hipMalloc(&c->rot_res, sizeof(SurfaceResult) * max_batch));
    TRY_C(hipMalloc(&c->trans_res, sizeof(SurfaceResult) * c->max_items));
    TRY_C(hipMalloc(&c->d_idx, sizeof(int) * c->max_items * IX_COUNT));
    TRY_C(hipHostMalloc(&c->h_idx, sizeof(int) * c->max_items * IX_COUNT));
    TRY_C(hipHostMalloc(&c->h_rot, sizeof(SurfaceResult) * max_batch));
    TRY_C(hipHostMalloc(&c->h_trans, sizeof(SurfaceResult) * c->max_items));
    TRY_C(hipMalloc(&c->d_u8, (size_t)H * W));
    TRY_C(hipMalloc(&c->d_scratch, sizeof(float) * std::max(c->img.real_elems, 2 * c->spec_max) * 2));
    if ((rc = build_polar_table(c)) || (rc = build_rot_table(c))) return bail(rc);
    TRY_C(hipStreamSynchronize(c->stream));
#undef TRY_C
    *out = c;
    return NIK_OK;
}

void nik_destroy(nik_ctx* c) {
    if (!c) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (Family* f : { &c->img, &c->pol }) for (float2* p : f->d_tw) (void)hipFree(p);
    (void)hipFree(c->arena_img); (void)hipFree(c->arena_F); (void)hipFree(c->arena_P);
    (void)hipFree(c->tmpA); (void)hipFree(c->kbuf); (void)hipFree(c->gbuf); (void)hipFree(c->splane); (void)hipFree(c->partials);
    (void)hipFree(c->maxbuf); (void)hipFree(c->energy); (void)hipFree(c->rot_res); (void)hipFree(c->trans_res); (void)hipFree(c->d_idx);
    if (c->h_idx) (void)hipHostFree(c->h_idx);
    if (c->h_rot) (void)hipHostFree(c->h_rot);
    if (c->h_trans) (void)hipHostFree(c->h_trans);
    (void)hipFree(c->d_u8); (void)hipFree(c->d_scratch); (void)hipFree(c->polar_tab); (void)hipFree(c->rot_tab);
    for (auto& r : c->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : c->prof_pool) (void)hipEventDestroy(e);
    if (c->idx_event) (void)hipEventDestroy(c->idx_event);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* nik_last_error(const nik_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int nik_get_dims(const nik_ctx* c, int dims[6]) {
    if (!c || !dims) return NIK_ERR_INVALID_ARG;
    dims[0] = c->H; dims[1] = c->W; dims[2] = c->PD; dims[3] = c->PC; dims[4] = c->max_batch; dims[5] = c->max_frames;
    return NIK_OK;
}
void* nik_stream(const nik_ctx* c) { return c ? (void*)c->stream : nullptr; }

int nik_synchronize(nik_ctx* c) {
    if (!c) return NIK_ERR_INVALID_ARG;
    int rc = drain_pending(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NIK_OK;
}

int nik_intermedium_batch_dev(nik_ctx* c, int n, const uint8_t* d_gray, const nik_frame* dst) {
    if (!c || !d_gray || !dst || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    if (n == 0) return NIK_OK;
    if (n > c->max_batch) return fail(c, NIK_ERR_CAPACITY, "batch %d exceeds max_batch %d", n, c->max_batch);
    int rc;
    if ((rc = drain_pending(c)) || (rc = begin_idx(c))) return rc;
    for (int i = 0; i < n; ++i) { if ((rc = check_slot(c, dst[i], false))) return rc; hidx(c, IX_DST)[i] = dst[i]; }
    if ((rc = upload_idx(c, IX_DST, n))) return rc;
    { Stage st(c, "k_cvt_u8", n * (1.0 * c->img.real_elems + Rb(c->img)));
      launch_cvt_u8(c->stream, n, d_gray, didx(c, IX_DST), c->arena_img, c->H, c->W); }
    enqueue_intermedium(c, n);
    HIP_TRY(c, hipGetLastError());
    for (int i = 0; i < n; ++i) c->slot_ready[dst[i]] = 3;
    return NIK_OK;
}

int nik_intermedium_u8(nik_ctx* c, const uint8_t* gray, int stride, nik_frame dst) {
    if (!c || !gray) return fail(c, NIK_ERR_INVALID_ARG, "null argument");
    if (stride < c->W) return fail(c, NIK_ERR_INVALID_ARG, "stride %d smaller than width %d", stride, c->W);
    int rc;
    if ((rc = drain_pending(c))) return rc;
    HIP_TRY(c, hipMemcpy2DAsync(c->d_u8, c->W, gray, stride, c->W, c->H, hipMemcpyHostToDevice, c->stream));
    rc = nik_intermedium_batch_dev(c, 1, c->d_u8, &dst);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NIK_OK;
}

int nik_intermedium_f32(nik_ctx* c, const float* image, nik_frame dst) {
    if (!c || !image) return fail(c, NIK_ERR_INVALID_ARG, "null argument");
    int rc;
    if ((rc = drain_pending(c)) || (rc = begin_idx(c)) || (rc = check_slot(c, dst, false))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->arena_img + (size_t)dst * c->img.real_elems, image, sizeof(float) * c->img.real_elems,
                              hipMemcpyHostToDevice, c->stream));
    hidx(c, IX_DST)[0] = dst;
    if ((rc = upload_idx(c, IX_DST, 1))) return rc;
    enqueue_intermedium(c, 1);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->slot_ready[dst] = 3;
    return NIK_OK;
}

int nik_frame_export(nik_ctx* c, nik_frame f, float* image, float* fft_result, float* fft_polar) {
    if (!c) return NIK_ERR_INVALID_ARG;
    int rc;
    if ((rc = drain_pending(c)) || (rc = check_slot(c, f, true))) return rc;
    hipStream_t s = c->stream;
    if (image) HIP_TRY(c, hipMemcpyAsync(image, c->arena_img + (size_t)f * c->img.real_elems, sizeof(float) * c->img.real_elems, hipMemcpyDeviceToHost, s));
    float2* scratch = reinterpret_cast<float2*>(c->d_scratch);
    if (fft_result) {     // internal [hr][W] -> reference column-major (hr x W) == [W][hr]
        launch_transpose_c(s, c->arena_F + (size_t)f * c->img.spec_elems, scratch, c->img.g.hr, c->W);
        HIP_TRY(c, hipMemcpyAsync(fft_result, scratch, sizeof(float2) * c->img.spec_elems, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    if (fft_polar) {
        launch_transpose_c(s, c->arena_P + (size_t)f * c->pol.spec_elems, scratch, c->pol.g.hr, c->PC);
        HIP_TRY(c, hipMemcpyAsync(fft_polar, scratch, sizeof(float2) * c->pol.spec_elems, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    return NIK_OK;
}

int nik_frame_import(nik_ctx* c, nik_frame f, const float* image, const float* fft_result, const float* fft_polar) {
    if (!c) return NIK_ERR_INVALID_ARG;
    int rc;
    if ((rc = drain_pending(c)) || (rc = check_slot(c, f, false))) return rc;
    hipStream_t s = c->stream;
    float2* scratch = reinterpret_cast<float2*>(c->d_scratch);
    if (image) { HIP_TRY(c, hipMemcpyAsync(c->arena_img + (size_t)f * c->img.real_elems, image, sizeof(float) * c->img.real_elems, hipMemcpyHostToDevice, s)); c->slot_ready[f] |= 1; }
    if (fft_result) {
        HIP_TRY(c, hipMemcpyAsync(scratch, fft_result, sizeof(float2) * c->img.spec_elems, hipMemcpyHostToDevice, s));
        launch_transpose_c(s, scratch, c->arena_F + (size_t)f * c->img.spec_elems, c->W, c->img.g.hr);
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    if (fft_polar) {
        HIP_TRY(c, hipMemcpyAsync(scratch, fft_polar, sizeof(float2) * c->pol.spec_elems, hipMemcpyHostToDevice, s));
        launch_transpose_c(s, scratch, c->arena_P + (size_t)f * c->pol.spec_elems, c->PC, c->pol.g.hr);
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    if (fft_result && fft_polar) c->slot_ready[f] |= 2;
    return NIK_OK;
}

int nik_pose_batch(nik_ctx* c, int n, const nik_frame* keys, const nik_frame* curs, int not_large_rotation,
                   nik_pose_result* res) {
    if (!c || !keys || !curs || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    int rc;
    if ((rc = check_kernel(c))) return rc;
    if (n == 0) return NIK_OK;
    if (n > c->max_batch) return fail(c, NIK_ERR_CAPACITY, "batch %d exceeds max_batch %d", n, c->max_batch);
    if ((rc = drain_pending(c)) || (rc = begin_idx(c))) return rc;
    for (int i = 0; i < n; ++i) {
        if ((rc = check_slot(c, keys[i], true)) || (rc = check_slot(c, curs[i], true))) return rc;
        hidx(c, IX_KEY)[i] = keys[i]; hidx(c, IX_CUR)[i] = curs[i];
    }
    if ((rc = upload_idx(c, IX_KEY, n)) || (rc = upload_idx(c, IX_CUR, n))) return rc;
    if ((rc = enqueue_pose(c, n, not_large_rotation))) return rc;
    HIP_TRY(c, hipGetLastError());
    c->pending.active = true; c->pending.n = n; c->pending.n_hyp = not_large_rotation ? 1 : 2; c->pending.res = res;
    return drain_pending(c);
}

int nik_pose(nik_ctx* c, nik_frame key, nik_frame cur, int not_large_rotation, double pose[3], double info[3],
             nik_pose_result* res) {
    nik_pose_result r;
    int rc = nik_pose_batch(c, 1, &key, &cur, not_large_rotation, &r);
    if (rc) return rc;
    if (pose) memcpy(pose, r.pose, sizeof(r.pose));
    if (info) memcpy(info, r.info, sizeof(r.info));
    if (res) *res = r;
    return NIK_OK;
}

int nik_track_batch_dev(nik_ctx* c, int n, const uint8_t* d_gray, const nik_frame* keys, const nik_frame* cur_dst,
                        int not_large_rotation, nik_pose_result* res, int sync) {
    if (!c || !d_gray || !keys || !cur_dst || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    int rc;
    if ((rc = check_kernel(c))) return rc;
    if (n == 0) return NIK_OK;
    if (n > c->max_batch) return fail(c, NIK_ERR_CAPACITY, "batch %d exceeds max_batch %d", n, c->max_batch);
    if ((rc = drain_pending(c)) || (rc = begin_idx(c))) return rc;
    for (int i = 0; i < n; ++i) {
        if ((rc = check_slot(c, keys[i], true)) || (rc = check_slot(c, cur_dst[i], false))) return rc;
        hidx(c, IX_KEY)[i] = keys[i]; hidx(c, IX_CUR)[i] = cur_dst[i]; hidx(c, IX_DST)[i] = cur_dst[i];
    }
    if ((rc = upload_idx(c, IX_KEY, n)) || (rc = upload_idx(c, IX_CUR, n)) || (rc = upload_idx(c, IX_DST, n))) return rc;
    { Stage st(c, "k_cvt_u8", n * (1.0 * c->img.real_elems + Rb(c->img)));
      launch_cvt_u8(c->stream, n, d_gray, didx(c, IX_DST), c->arena_img, c->H, c->W); }
    enqueue_intermedium(c, n);
    for (int i = 0; i < n; ++i) c->slot_ready[cur_dst[i]] = 3;
    if ((rc = enqueue_pose(c, n, not_large_rotation))) return rc;
    HIP_TRY(c, hipGetLastError());
    c->pending.active = true; c->pending.n = n; c->pending.n_hyp = not_large_rotation ? 1 : 2; c->pending.res = res;
    return sync ? drain_pending(c) : NIK_OK;
}

int nik_match(nik_ctx* c, nik_frame query, int n, const nik_frame* cands, int* best, nik_pose_result* res,
              nik_pose_result* best_res) {
    if (!c || (n > 0 && !cands) || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    if (best) *best = -1;
    if (n == 0) return NIK_OK;
    std::vector<nik_frame> curs(n, query);
    std::vector<nik_pose_result> local;
    if (!res) { local.resize(n); res = local.data(); }
    int rc = nik_pose_batch(c, n, cands, curs.data(), 0, res);       // loop_closure.cc:58-59 (not_large_rotation=false)
    if (rc) return rc;
    int b = -1; double bs = -3.0;                                    // LoopClosureResult(): response(-1,-1,-1)  (loop_closure.h:14)
    for (int i = 0; i < n; ++i) {
        const double s = res[i].info[0] + res[i].info[1] + res[i].info[2];
        if (s > bs) { bs = s; b = i; }                               // loop_closure.cc:61 strict >
    }
    if (best) *best = b;
    if (best_res && b >= 0) *best_res = res[b];
    return NIK_OK;
}

int nik_profile_enable(nik_ctx* c, int enable) {
    if (!c) return NIK_ERR_INVALID_ARG;
    int rc = drain_pending(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (auto& r : c->prof_recs) { c->prof_pool.push_back(r.a); c->prof_pool.push_back(r.b); }
    c->prof_recs.clear(); c->prof_stats.clear();
    c->prof_on = enable != 0;
    return NIK_OK;
}

int nik_profile_read(nik_ctx* c, nik_stage_stat* out, int cap, int* n) {
    if (!c || !n) return NIK_ERR_INVALID_ARG;
    int rc = drain_pending(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (auto& r : c->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) c->prof_stats[r.stage].ms += ms;
        c->prof_pool.push_back(r.a); c->prof_pool.push_back(r.b);
    }
    c->prof_recs.clear();
    *n = (int)c->prof_stats.size();
    for (int i = 0; i < *n && i < cap && out; ++i) {
        memset(&out[i], 0, sizeof(out[i]));
        strncpy(out[i].name, c->prof_stats[i].name.c_str(), sizeof(out[i].name) - 1);
        out[i].ms = c->prof_stats[i].ms; out[i].launches = c->prof_stats[i].launches; out[i].bytes = c->prof_stats[i].bytes;
    }
    return NIK_OK;
}

// ---- debug taps ---------------------------------------------------------------------------------

int nik_dbg_set_ablate(int flags) { set_ablate(flags); return NIK_OK; }

int nik_dbg_fft(nik_ctx* c, int which, const float* x, float* xf_out) {
    if (!c || !x || !xf_out) return NIK_ERR_INVALID_ARG;
    int rc;
    if ((rc = drain_pending(c))) return rc;
    Family& f = which ? c->pol : c->img;
    hipStream_t s = c->stream;
    float* d_in = c->d_scratch;                                                   // first half: real input
    float2* d_out = reinterpret_cast<float2*>(c->d_scratch) + c->spec_max;        // second half: transposed output
    HIP_TRY(c, hipMemcpyAsync(d_in, x, sizeof(float) * f.real_elems, hipMemcpyHostToDevice, s));
    launch_A_fwd_plane(s, 1, f.g, f.t, d_in, f.real_elems, nullptr, c->tmpA, c->spec_max);
    launch_B_fwd(s, 1, f.g, f.t, c->tmpA, c->spec_max, c->tmpA, c->spec_max, nullptr);
    launch_transpose_c(s, c->tmpA, d_out, f.g.hr, f.g.cols);
    HIP_TRY(c, hipMemcpyAsync(xf_out, d_out, sizeof(float2) * f.spec_elems, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    return NIK_OK;
}

int nik_dbg_ifft(nik_ctx* c, int which, const float* xf, float* x_out) {
    if (!c || !xf || !x_out) return NIK_ERR_INVALID_ARG;
    int rc;
    if ((rc = drain_pending(c))) return rc;
    Family& f = which ? c->pol : c->img;
    hipStream_t s = c->stream;
    float2* scratch = reinterpret_cast<float2*>(c->d_scratch);
    HIP_TRY(c, hipMemcpyAsync(scratch, xf, sizeof(float2) * f.spec_elems, hipMemcpyHostToDevice, s));
    launch_transpose_c(s, scratch, c->tmpA, f.g.cols, f.g.hr);
    launch_B_inv(s, 1, f.g, f.t, c->tmpA, c->spec_max, c->gbuf, c->spec_max);
    float* dst = reinterpret_cast<float*>(c->kbuf);
    launch_A_inv_real(s, 1, f.g, f.t, c->gbuf, c->spec_max, dst, f.real_elems);
    HIP_TRY(c, hipMemcpyAsync(x_out, dst, sizeof(float) * f.real_elems, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    return NIK_OK;
}

int nik_dbg_rotate(nik_ctx* c, nik_frame fr, int degree2, float* out) {
    if (!c || !out) return NIK_ERR_INVALID_ARG;
    int rc;
    if ((rc = drain_pending(c)) || (rc = check_slot(c, fr, false))) return rc;
    if (!(c->slot_ready[fr] & 1)) return fail(c, NIK_ERR_NOT_READY, "frame slot %d holds no image", fr);
    hipStream_t s = c->stream;
    std::vector<int> terms((size_t)2 * c->W + 2 * c->H);
    rotation_terms(c->H, c->W, (float)degree2 * 0.5f, terms.data());             // RotateArray(image, degree2/2)
    int* d_terms = reinterpret_cast<int*>(c->gbuf);
    HIP_TRY(c, hipMemcpyAsync(d_terms, terms.data(), sizeof(int) * terms.size(), hipMemcpyHostToDevice, s));
    launch_dbg_rot(s, c->arena_img + (size_t)fr * c->img.real_elems, d_terms, c->d_scratch, c->H, c->W);
    HIP_TRY(c, hipMemcpyAsync(out, c->d_scratch, sizeof(float) * c->img.real_elems, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    return NIK_OK;
}

int nik_dbg_polar(nik_ctx* c, const float* x, float* out) {
    if (!c || !x || !out) return NIK_ERR_INVALID_ARG;
    int rc;
    if ((rc = drain_pending(c))) return rc;
    hipStream_t s = c->stream;
    float* d_out = reinterpret_cast<float*>(c->gbuf);
    float* d_in = c->d_scratch;
    HIP_TRY(c, hipMemcpyAsync(d_in, x, sizeof(float) * c->img.real_elems, hipMemcpyHostToDevice, s));
    launch_make_shifted(s, d_in, c->splane, c->H, c->W);
    launch_fix_zero(s, 1, c->splane, c->s_elems, c->H, c->W);
    launch_dbg_polar(s, c->splane, c->polar_tab, d_out, c->H, c->W, c->PD, c->PC);
    HIP_TRY(c, hipMemcpyAsync(out, d_out, sizeof(float) * c->pol.real_elems, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    return NIK_OK;
}

}  // extern "C"
