// kcc_api.hip -- C ABI of libnislam_kcc_hip.so (see include/nislam_kcc.h).
// Host side: context, device-resident keyframe store, batched stage scheduling.
// Mirrors CorrelationFlow (reference include/correlation_flow.h:8-33, src/correlation_flow.cc:37-143).
//
// Scheduling: a context owns NL "lanes" (HIP stream + private work buffers).  A batched call is split into
// NL contiguous chunks, one per lane, so the kernels of different chunks overlap on the GPU (the memory-bound
// phases of one chunk hide under the ALU/LDS-bound phases of another).  Each lane keeps a ring of two
// in-flight calls (pinned index / result staging + a completion event), so the host can queue the next batch
// while the previous one is still running; results are finalised lazily (nik_synchronize or ring reuse).
#include "../../include/nislam_kcc.h"
#include "kcc_kernels.h"
#include "kcc_generic.h"
#include "kcc_tables.h"
#include "kcc_tune.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace kcc;

enum { KCC_RING_MAX = 4 };
enum { KCC_STATS_PARTS = 4096 };   // chunk partials of the residual statistics one API call may produce

namespace {

thread_local std::string g_create_error;

struct Family {                 // one plane geometry with its tables
    PlaneGeom g{};
    Tables t{};
    size_t real_elems = 0;      // rows*cols
    size_t spec_elems = 0;      // hr*cols
    float2* d_tw[9] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
};

// index arrays of one call (each cap_items ints)
enum { IX_KEY = 0, IX_CUR = 1, IX_DST = 2, IX_TIMG = 3, IX_TKEY = 4, IX_ROTIDX = 5,
       IX_WRR = 6, IX_WRC = 7, IX_WTR = 8, IX_WTC = 9,       // arg-max window centres (rotation / translation surface), host-written
       IX_CVT = 10,                                           // u8 frames to materialise as f32 planes (mixed u8 / f32 batches)
       IX_COUNT = 11 };

struct Call {                   // one in-flight call on a lane
    int* h_idx = nullptr;                                   // pinned staging, IX_COUNT * cap_items
    SurfaceResult* h_rot = nullptr; SurfaceResult* h_trans = nullptr;   // pinned result staging
    hipEvent_t done = nullptr;
    bool busy = false;          // `done` has been recorded and not yet waited for
    bool has_pose = false; int n = 0, n_hyp = 1; nik_pose_result* res = nullptr;
};

struct Window { const int* row = nullptr; const int* col = nullptr; int radius = 0, mirror = 0; };   // arg-max window (device arrays)

struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t write_ev = nullptr;      // recorded after the lane's latest frame-slot writes
    unsigned long write_seq = 0;
    std::vector<unsigned long> seen;    // seen[w] = write_seq of lane w this lane has already waited for
    hipEvent_t tail_ev = nullptr;       // recorded at the end of the lane's latest call
    int flip = 0;                       // item order of the lane's next big kernel (alternates: see Stage)
    unsigned long call_seq = 0;
    std::vector<unsigned long> seen_tail;   // seen_tail[r] = call_seq of lane r this lane has already waited for
    int cap_items = 0;
    float2* tmpA = nullptr;             // [cap_items][max spec]
    float2* kbuf = nullptr;             // [cap_items][2][max spec]  (zz, xz planes)
    float2* gbuf = nullptr;             // [cap_items][max spec]
    void* slab = nullptr;               // tuning: tmpA | kbuf | gbuf as one allocation ($NIK_LANE_SLAB)
    float*  splane = nullptr;           // [cap_pairs][(W+1)*(H+2)] shifted zero-bordered planes (polar source)
    uint8_t* u8tmp = nullptr;           // [cap_pairs][H*W] undistorted frames (allocated by nik_set_undistort)
    float*  rbuf = nullptr;             // generic-size contexts: [cap_items][2][max real plane] real work planes
    Partial* partials = nullptr;
    unsigned* maxbuf = nullptr; float* energy = nullptr;
    std::vector<int> idx_shadow; int idx_shadow_n[5] = { 0, 0, 0, 0, 0 }; bool idx_force = false;   // host mirror of d_idx[0, IX_ROTIDX) and the valid prefix of each array
    SurfaceResult* rot_res = nullptr; SurfaceResult* trans_res = nullptr;
    int* d_idx = nullptr;
    Call ring[KCC_RING_MAX]; int ring_n = 2, next = 0;   // ring_n calls in flight before the host blocks on the oldest (nik_set_call_depth)
    Call* cur = nullptr;                // call being enqueued
    SurfaceResult* mirror = nullptr;    // pinned host mirror for the results of the next enqueue_estimate (consumed by it)
    // hipGraph replay of the pose chain for small batches (nik_set_graphs): one executable per (ring entry, n, mode)
    struct PoseGraph { int slot, n, flags; hipGraphExec_t exec; int uses; };
    std::vector<PoseGraph> graphs;
};

}  // namespace

struct nik_ctx {
    nik_config cfg{};
    int H = 0, W = 0, PD = 0, PC = 0;
    int max_batch = 0, max_frames = 0, device = 0;
    int max_items = 0;          // 2*max_batch (two hypotheses per pair in large-rotation mode)
    std::string err;

    Family img, pol;
    // any-size fallback (kcc_generic.hip): set when the geometry is outside the tiled kernels' instantiated set (or $NIK_GENERIC=1)
    bool generic = false;          // either family below is on the any-size kernels (work planes allocated; no graphs / Kzz cache / deferred passes)
    bool gen_img = false, gen_pol = false;   // per plane family: a 640x480 camera with a 720x64 polar plane keeps the tiled image kernels
    GFamily gimg{}, gpol{}; float2* g_tw[4] = { nullptr, nullptr, nullptr, nullptr }; uint32_t* g_polar_map = nullptr;
    // keyframe store (reference Frame: _frame, _fft_result, _fft_polar)
    // The image of a frame lives as u8 (row-major, what the u8 entry points receive) or as f32 (column-major, what the
    // reference's ArrayXXf entry points hand over); slot_kind says which copies are valid.
    uint8_t* arena_u8 = nullptr; float* arena_img = nullptr; float2* arena_F = nullptr; float2* arena_P = nullptr;
    int u8_pitch = 0; size_t u8_stride = 0;     // u8 images: row pitch W + 16 (columns 0..15 repeated behind column W-1: BORDER_WRAP), bytes per slot
    int img_pitch = 0; size_t img_stride = 0;   // f32 planes: column pitch >= H + 4 (rows H..H+3 repeat rows 0..3), elements per slot
    std::vector<uint8_t> slot_ready;     // bit0: image, bit1: spectra
    std::vector<uint8_t> slot_kind;      // bit0: u8 image valid, bit1: f32 image valid
    // optional per-keyframe Kzz cache (SURVEY 8d "Kzz cached"): transformed kernel spectrum + max per slot and family
    bool kzz_cache = false;
    int16_t* ud_map1 = nullptr; uint16_t* ud_map2 = nullptr;   // undistortion maps (nik_set_undistort); null = u8 inputs are already undistorted
    bool zz_half = true;          // uncached Kzz: transform only the Hermitian half of its kernel plane ($NIK_ZZ_HALF=0: off)
    bool fuse_polar = true;       // tracking path: fuse the polar spectrum's last pass into the pose's first kernel ($NIK_FUSE_POLAR=0: off)
    bool fuse_fix_zero = true;    // RemoveZeroComponent inside the shifted inverse kernel ($NIK_FUSE_FIX_ZERO=0: its own launch)
    bool alt_order = true;        // consecutive kernels of a lane walk the items in opposite directions ($NIK_ALT_ORDER=0: off)
    int chunk_pairs = 0;          // batched calls are cut into chunks of at most this many pairs, dealt to the lanes in turn (0: one chunk per lane; $NIK_CHUNK)
    float2* arena_KzF = nullptr; float2* arena_KzP = nullptr; unsigned* arena_MzF = nullptr; unsigned* arena_MzP = nullptr;
    std::vector<uint8_t> slot_kzz;       // 1: cache valid
    std::vector<int8_t> slot_lane;       // lane that last wrote the slot (-1: none / host import)
    std::vector<unsigned long> slot_seq; // that lane's write_seq at the time
    std::vector<unsigned long> slot_rd;  // [slot][4]: call_seq of each lane's latest call that read the slot
    size_t s_elems = 0, spec_max = 0, r_elems = 0; int partial_stride = 0;
    std::vector<Lane> lanes; int active_lanes = 1;
    uint8_t* d_u8 = nullptr;             // staging for host u8 input (one image)
    // host frames -> device on the context's own upload stream (nik_upload_u8_async): never on a compute lane
    hipStream_t up_stream = nullptr; hipEvent_t up_ev[4] = { nullptr, nullptr, nullptr, nullptr }; unsigned up_seq = 0; int up_fenced = -1; hipEvent_t up_after_ev = nullptr;
    uint8_t* up_pin[2] = { nullptr, nullptr }; hipEvent_t up_pin_ev[2] = { nullptr, nullptr }; bool up_pin_busy[2] = { false, false };
    size_t up_pin_bytes = 0; int up_pin_next = 0;
    float* d_scratch = nullptr;          // debug / import-export staging
    // residual statistics of the latest batch (nik_set_residual_stats): per-lane partials [4 lanes][4] + their sum [4], device
    // They are summed (and all-reduced, nik_group) on their own stream so that no lane waits for another.
    bool want_stats = false; double* d_stats = nullptr; double* h_stats = nullptr; int stats_lanes = 0;
    int stats_parts = 0;                 // partial blocks [4] written by the chunks of the current API call (reset at API entry)
    int last_pose_n = -1, last_pose_nhyp = 0, last_pose_nc = 0;   // shape of the latest pose call (nik_pose_batch_chained checks it)
    hipStream_t stats_stream = nullptr; hipEvent_t stats_done = nullptr; bool stats_pending = false;
    int graph_max = 0;                   // batches of <= graph_max pairs replay a captured hipGraph (0: off); $NIK_GRAPH
    int lane_items = 32;                 // a call of n items is spread over n / lane_items streams ($NIK_LANE_ITEMS)
    bool lane_rot = false; int lane_base = 0;   // nik_set_lane_rotation: successive calls start on successive lanes (several small calls in flight run side by side)
    hipEvent_t fence_ev = nullptr;       // nik_wait_for
    hipEvent_t chain_ev[4] = { nullptr, nullptr, nullptr, nullptr };   // a finer pyramid level has read lane li's surface results
    bool chain_pending[4] = { false, false, false, false };
    PolarPlan polar{};                   // gather tables of the polar forward kernel (device pointers; kcc_tables.cpp)
    int* rot_one = nullptr;              // one-angle de-rotation table (nik_dbg_rotate)
    int* rot_tab = nullptr;              // [3][PD][2W+2H] fixed-point warpAffine terms per candidate angle
    std::vector<float> rot_deg;          // [3][PD] degree after normalise/fold (variant 0) or hypothesis angles
    // per-stage HIP-event profiler (nik_profile_enable / nik_profile_read)
    struct StageStat { std::string name; double ms = 0; long launches = 0; double bytes = 0, bytes_design = 0; };
    struct StageRec { int stage; hipEvent_t a, b; };
    bool prof_on = false;
    std::vector<StageStat> prof_stats;
    std::vector<StageRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
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
        // 16 x PR plans (kcc_fft2.h PlanPrime): W_PR^(-+ m), m < PR, behind the pass table
        for (int m = 0; m < d.prime; ++m) t.push_back(w(m, d.prime));
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
    if ((&f == &c->img) ? c->gen_img : c->gen_pol) return NIK_OK;   // (run-time plans: generic_init)
    if (kfwd_parts(f.g, false) > KCC_MAXPARTS) return fail(c, NIK_ERR_UNSUPPORTED_SIZE, "%d x %d: more running-max parts than KCC_MAXPARTS", rows, cols);
    const int h = rows / 2;
    const PlanDesc ph = plan_desc(h), pc = plan_desc(cols);
    int rc;
    if ((rc = upload_table(c, plan_table(ph, false), &f.d_tw[0])) || (rc = upload_table(c, plan_table(ph, true), &f.d_tw[1])) ||
        (rc = upload_table(c, twiddles(rows, h), &f.d_tw[2])) ||
        (rc = upload_table(c, plan_table(pc, false), &f.d_tw[3])) || (rc = upload_table(c, plan_table(pc, true), &f.d_tw[4])) ||
        (rc = upload_table(c, plan_table(plan_desc_inv(h), false), &f.d_tw[5])) || (rc = upload_table(c, plan_table(plan_desc_inv(h), true), &f.d_tw[6])) ||
        (rc = upload_table(c, plan_table(plan_desc_alt(cols), false), &f.d_tw[7])) || (rc = upload_table(c, plan_table(plan_desc_alt(cols), true), &f.d_tw[8]))) return rc;
    f.t.colsA_f = f.d_tw[7]; f.t.colsA_i = f.d_tw[8];
    f.t.halfI_f = f.d_tw[5]; f.t.halfI_i = f.d_tw[6];
    f.t.half_f = f.d_tw[0]; f.t.half_i = f.d_tw[1]; f.t.tw_full = f.d_tw[2]; f.t.cols_f = f.d_tw[3]; f.t.cols_i = f.d_tw[4];
    return NIK_OK;
}

// any-size contexts: run-time FFT plans (one exp(-2 pi i k / n) table per line length, evaluated in double) and the per-pixel
// polar map (the same build_polar_map the tiled kernel's plan is derived from)
int generic_init(nik_ctx* c) {
    const int len[4] = { c->H, c->W, c->PD, c->PC };
    for (int q = 0; q < 4; ++q) { int rc = upload_table(c, twiddles(len[q], len[q]), &c->g_tw[q]); if (rc) return rc; }
    c->gimg.g = c->img.g; c->gimg.prow = gplan_make(c->H, c->g_tw[0]); c->gimg.pcol = gplan_make(c->W, c->g_tw[1]);
    c->gpol.g = c->pol.g; c->gpol.prow = gplan_make(c->PD, c->g_tw[2]); c->gpol.pcol = gplan_make(c->PC, c->g_tw[3]);
    for (const GPlan* p : { &c->gimg.prow, &c->gimg.pcol, &c->gpol.prow, &c->gpol.pcol })
        if (p->nr == 0) return fail(c, NIK_ERR_UNSUPPORTED_SIZE, "FFT length %d has more prime factors than the run-time plan holds", p->n);
    if (!c->gen_pol) return NIK_OK;
    std::vector<uint32_t> map; std::string err;
    if (build_polar_map(c->H, c->W, c->PD, c->PC, map, err)) return fail(c, NIK_ERR_UNSUPPORTED_SIZE, "%s", err.c_str());
    HIP_TRY(c, hipMalloc(&c->g_polar_map, sizeof(uint32_t) * map.size()));
    HIP_TRY(c, hipMemcpy(c->g_polar_map, map.data(), sizeof(uint32_t) * map.size(), hipMemcpyHostToDevice));
    return NIK_OK;
}

// upload the polar gather plan (kcc_tables.cpp) for the forward kernel's tile geometry
int build_polar_table(nik_ctx* c) {
    const FwdGeom fg = fwd_geom(c->PD / 2);
    PolarPlanHost h; std::string err;
    const char* al = kcc::tune_env("NIK_POLAR_ALIGNED");
    if (build_polar_plan(c->H, c->W, c->PD, c->PC, fg.lines, fg.threads, fg.rf, fg.mf, fg.lds_bytes, fg.qs_opts, h, err, al ? atoi(al) : KCC_POLAR_ALIGNED_DEFAULT))
        return fail(c, NIK_ERR_UNSUPPORTED_SIZE, "%s", err.c_str());
    uint32_t* d_chunks = nullptr; int* d_first = nullptr; uint4* d_pts = nullptr;
    HIP_TRY(c, hipMalloc(&d_chunks, sizeof(uint32_t) * std::max<size_t>(h.chunks.size(), 1)));
    c->polar.chunks = d_chunks;
    HIP_TRY(c, hipMalloc(&d_first, sizeof(int) * h.seg_first.size()));
    c->polar.seg_first = d_first;
    HIP_TRY(c, hipMalloc(&d_pts, sizeof(uint32_t) * h.pts.size()));
    c->polar.pts = d_pts;
    HIP_TRY(c, hipMemcpy(d_chunks, h.chunks.data(), sizeof(uint32_t) * h.chunks.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(d_first, h.seg_first.data(), sizeof(int) * h.seg_first.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(d_pts, h.pts.data(), sizeof(uint32_t) * h.pts.size(), hipMemcpyHostToDevice));
    c->polar.qs = h.qs; c->polar.lds_bytes = h.lds_bytes;
    return NIK_OK;
}

double normalize_degree(double a) { return a - 360 * floor((a + 180) / 360); }     // utils.cc:173-175

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

inline int* didx(Lane& L, int which) { return L.d_idx + (size_t)which * L.cap_items; }
inline int* hidx(Lane& L, int which) { return L.cur->h_idx + (size_t)which * L.cap_items; }

// The device copy of an index array is left alone when it already holds these values (a tracker or a pyramid level that
// works on the same slots call after call): one stream operation fewer per call, which is what small batches are bound by.
// L.idx_shadow mirrors what the host has uploaded into d_idx[0, IX_ROTIDX) (the arrays behind are written by kernels).
static_assert(IX_ROTIDX == 5, "Lane::idx_shadow_n and stage_pose_indices::used list the five host-filled index arrays");
inline bool idx_unchanged(Lane& L, int which, int n) {
    return n <= L.idx_shadow_n[which] && memcmp(L.idx_shadow.data() + (size_t)which * L.cap_items, hidx(L, which), sizeof(int) * n) == 0;
}
inline void idx_remember(Lane& L, int which, int n) {
    memcpy(L.idx_shadow.data() + (size_t)which * L.cap_items, hidx(L, which), sizeof(int) * n);
    L.idx_shadow_n[which] = n;
}
int upload_idx(nik_ctx* c, Lane& L, int which, int n) {
    if (which < IX_ROTIDX && idx_unchanged(L, which, n)) return NIK_OK;
    HIP_TRY(c, hipMemcpyAsync(didx(L, which), hidx(L, which), sizeof(int) * n, hipMemcpyHostToDevice, L.stream));
    if (which < IX_ROTIDX) idx_remember(L, which, n);
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

// GetInfo (correlation_flow.cc:238-243) from single-pass moments
float psr_from(const SurfaceResult& r, long n) {
    const double m = (r.sum - (double)r.peak) / (double)(n - 1);
    double var = (r.sumsq - 2.0 * m * r.sum + (double)n * m * m) / (double)n;
    if (var < 0) var = 0;
    return (float)(((double)r.peak - m) / ((double)(float)sqrt(var) + 1e-7));
}

// host tail of ComputePose (:105-138) from the raw arg-max results of pair i of a call
void finalize_pose(nik_ctx* c, const Call& call, int i, nik_pose_result* out) {
    const int PD = c->PD, H = c->H, W = c->W, n_hyp = call.n_hyp;
    nik_pose_result r; memset(&r, 0, sizeof(r));
    const SurfaceResult& rr = call.h_rot[i];
    r.rot_row = rr.idx % PD; r.rot_col = rr.idx / PD;
    r.psr_rot = psr_from(rr, (long)PD * c->PC);
    r.n_hyp = n_hyp;
    float degree; float info_trans; double trans0, trans1;
    auto tr = [&](int hyp, double& t0, double& t1) {
        const SurfaceResult& s = call.h_trans[i * n_hyp + hyp];
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

// wait for one in-flight call and hand its results to the caller
int retire(nik_ctx* c, Call& call) {
    if (!call.busy) return NIK_OK;
    HIP_TRY(c, hipEventSynchronize(call.done));
    if (call.has_pose && call.res)
        for (int i = 0; i < call.n; ++i) finalize_pose(c, call, i, call.res + i);
    call.busy = false; call.has_pose = false; call.res = nullptr;
    return NIK_OK;
}

// Lanes beyond active_lanes never hold work (nik_set_streams drains before it changes the count): the stream plumbing below
// touches the active ones only -- every event call costs host time, and a chained pyramid issues dozens per batch
// (three allocated lanes instead of one cost the pyramid workload 30 %).
int drain_all(nik_ctx* c) {
    for (int li = 0; li < c->active_lanes; ++li) {
        Lane& L = c->lanes[li];
        for (int k = 0; k < L.ring_n; ++k) {        // oldest call first
            int rc = retire(c, L.ring[(L.next + k) % L.ring_n]);
            if (rc) return rc;
        }
    }
    return NIK_OK;
}

// start a call on a lane: take the older ring entry (retiring what it held)
int begin_call(nik_ctx* c, Lane& L) {
    Call& call = L.ring[L.next];
    int rc = retire(c, call);
    if (rc) return rc;
    L.cur = &call; L.next = (L.next + 1) % L.ring_n; L.call_seq += 1;
    const int li = (int)(&L - c->lanes.data());
    if (c->chain_pending[li]) { HIP_TRY(c, hipStreamWaitEvent(L.stream, c->chain_ev[li], 0)); c->chain_pending[li] = false; }
    return NIK_OK;
}
int end_call(nik_ctx* c, Lane& L) {
    HIP_TRY(c, hipEventRecord(L.cur->done, L.stream));
    HIP_TRY(c, hipEventRecord(L.tail_ev, L.stream));
    L.cur->busy = true;
    return NIK_OK;
}

// lane L is about to read frame slot f: order it after the lane that wrote f (if different and not yet seen)
int depend_on_slot(nik_ctx* c, Lane& L, int li, nik_frame f) {
    const int w = c->slot_lane[f];
    if (w < 0 || w == li) return NIK_OK;
    if (L.seen[w] >= c->slot_seq[f]) return NIK_OK;
    HIP_TRY(c, hipStreamWaitEvent(L.stream, c->lanes[w].write_ev, 0));
    L.seen[w] = c->lanes[w].write_seq;
    return NIK_OK;
}
// lane L reads slot f in its current call
inline void note_read(nik_ctx* c, Lane& L, int li, nik_frame f) { c->slot_rd[(size_t)f * 4 + li] = L.call_seq; }
// lane L is about to overwrite slot f: order it after other lanes' calls that read or wrote it
int depend_for_write(nik_ctx* c, Lane& L, int li, nik_frame f) {
    int rc = depend_on_slot(c, L, li, f);
    if (rc) return rc;
    for (int r = 0; r < (int)c->lanes.size(); ++r) {
        if (r == li) continue;
        if (c->slot_rd[(size_t)f * 4 + r] > L.seen_tail[r]) {
            HIP_TRY(c, hipStreamWaitEvent(L.stream, c->lanes[r].tail_ev, 0));
            L.seen_tail[r] = c->lanes[r].call_seq;
        }
    }
    return NIK_OK;
}
// kind: which image copy the call wrote (1 = u8, 2 = f32)
int mark_written(nik_ctx* c, Lane& L, int li, const int* slots, int n, int kind) {
    L.write_seq += 1;
    HIP_TRY(c, hipEventRecord(L.write_ev, L.stream));
    for (int i = 0; i < n; ++i) {
        c->slot_lane[slots[i]] = (int8_t)li; c->slot_seq[slots[i]] = L.write_seq; c->slot_ready[slots[i]] = 3; c->slot_kzz[slots[i]] = 0;
        c->slot_kind[slots[i]] = (uint8_t)kind;
    }
    return NIK_OK;
}

// f32 column-major planes (/255) for those of the listed slots that only hold a u8 image, on lane L's stream
int ensure_f32_images(nik_ctx* c, Lane& L, int li, int n, const nik_frame* slots) {
    int k = 0;
    for (int i = 0; i < n; ++i) {
        const nik_frame f = slots[i];
        if (c->slot_kind[f] & 2) continue;
        bool dup = false;
        for (int q = 0; q < k; ++q) dup |= hidx(L, IX_CVT)[q] == f;
        if (!dup) hidx(L, IX_CVT)[k++] = f;
    }
    if (!k) return NIK_OK;
    int rc = upload_idx(c, L, IX_CVT, k);
    if (rc) return rc;
    if (c->gen_img) g_cvt_u8(L.stream, k, c->arena_u8, c->u8_stride, c->u8_pitch, didx(L, IX_CVT), c->arena_img, c->H, c->W, c->img_pitch);
    else launch_cvt_u8(L.stream, k, c->arena_u8, c->u8_stride, c->u8_pitch, didx(L, IX_CVT), c->arena_img, c->H, c->W, c->img_pitch);
    // published as a slot write of this lane: other lanes order their reads of the new planes after it
    L.write_seq += 1;
    HIP_TRY(c, hipEventRecord(L.write_ev, L.stream));
    for (int i = 0; i < k; ++i) {
        const nik_frame f = hidx(L, IX_CVT)[i];
        c->slot_kind[f] |= 2; c->slot_lane[f] = (int8_t)li; c->slot_seq[f] = L.write_seq;
    }
    return NIK_OK;
}

// ---- stage profiler: brackets one kernel launch with HIP events on the launch stream -----------------
struct Stage {
    nik_ctx* c; hipStream_t s; int rec = -1;
    // bytes: nominal planes of the pass; design: what the launch is built to move (< 0: the same)
    Stage(nik_ctx* c_, Lane& L, const char* name, double bytes, double design = -1.0) : c(c_), s(L.stream) {
        // every big kernel of a lane consumes what the previous one produced: alternate the item order so that it starts
        // with the items written last (still in the Infinity Cache)
        if (c->alt_order) { set_launch_reverse(L.flip); L.flip ^= 1; }
        if (!c->prof_on) return;
        int id = -1;
        for (size_t i = 0; i < c->prof_stats.size(); ++i) if (c->prof_stats[i].name == name) { id = (int)i; break; }
        if (id < 0) { c->prof_stats.push_back({}); id = (int)c->prof_stats.size() - 1; c->prof_stats[id].name = name; }
        c->prof_stats[id].launches += 1; c->prof_stats[id].bytes += bytes; c->prof_stats[id].bytes_design += design < 0 ? bytes : design;
        nik_ctx::StageRec r; r.stage = id;
        for (hipEvent_t* e : { &r.a, &r.b }) {
            if (!c->prof_pool.empty()) { *e = c->prof_pool.back(); c->prof_pool.pop_back(); }
            else if (hipEventCreate(e) != hipSuccess) return;
        }
        (void)hipEventRecord(r.a, s);
        c->prof_recs.push_back(r); rec = (int)c->prof_recs.size() - 1;
    }
    ~Stage() { set_launch_reverse(0); if (rec >= 0) (void)hipEventRecord(c->prof_recs[rec].b, s); }
};
std::string kname(const char* base, int len, const char* mode, int other = 0) {
    // other > 0: the plane's second dimension, appended when the image and polar families share the kernel's template length
    // (1280x720 frames: both have 720 rows) so that their launches are not timed as one stage
    char b[64];
    if (other > 0) snprintf(b, sizeof(b), "%s<%d,%s>/%d", base, len, mode, other); else snprintf(b, sizeof(b), "%s<%d,%s>", base, len, mode);
    return b;
}
// second dimension to tag an A-type stage of family f with (0: no clash)
inline int a_tag(const nik_ctx* c, const Family& f) { return c->img.g.rows == c->pol.g.rows ? f.g.cols : 0; }
inline int b_tag(const nik_ctx* c, const Family& f) { return c->img.g.cols == c->pol.g.cols ? f.g.rows : 0; }
inline double Rb(const Family& f) { return 4.0 * (double)f.real_elems; }     // real plane bytes
inline double Cb(const Family& f) { return 8.0 * (double)f.spec_elems; }     // half-spectrum plane bytes

// ComputeIntermedium (correlation_flow.cc:89-95) for n frames whose spectra go to the arena slots listed in d_idx[IX_DST].
// d_u8 != null: the frames arrive as u8 row-major images (n contiguous); ConvertMatToNormalizedArray (utils.cc:110-118)
//   is fused into the first FFT pass, which also files the images in the u8 frame store.  With undistortion maps
//   installed d_u8 is the RAW camera frame and Camera::UndistortImage (camera.cc:92-93) runs first.
// d_u8 == null: the frames are f32 column-major planes already stored in the arena slots (nik_intermedium_f32).
// defer_polar_B: leave the polar spectrum's second (radius) pass to the caller -- the pose that follows fuses it into
// its first kernel (fwd_mul_inv), which also writes the finished spectrum to the frame store.  L.tmpA then holds the
// half-transformed polar spectra.
void enqueue_intermedium(nik_ctx* c, Lane& L, int n, const uint8_t* d_u8, bool defer_polar_B = false) {
    hipStream_t s = L.stream;
    const int* dst = didx(L, IX_DST);
    const Family& I = c->img; const Family& P = c->pol;
    const size_t RS = 2 * c->r_elems;                        // any-size kernels: real work planes per item, [2][r_elems]
    if (d_u8 && c->ud_map1) {
        Stage st(c, L, "k_undistort_u8", n * 2.0 * (double)I.real_elems + 6.0 * (double)I.real_elems);
        launch_undistort_u8(s, n, d_u8, L.u8tmp, c->ud_map1, c->ud_map2, c->H, c->W);
        d_u8 = L.u8tmp;
    }
    // ---- image family: fft_result, the zero-phase image, RemoveZeroComponent + fftshift -> the shifted plane S (:91-94)
    if (c->gen_img) {
        // any-size kernels (kcc_generic.hip): every stage its own launch, planes in the lane's work buffers
        if (d_u8) {
            { Stage st(c, L, "kg_u8_load", n * 5.0 * (double)I.real_elems);
              g_u8_load(s, n, d_u8, I.real_elems, L.rbuf, RS, c->arena_u8, c->u8_stride, c->u8_pitch, dst, c->H, c->W); }
            Stage st(c, L, "kg_rfft2<image>", n * (Rb(I) + 3 * Cb(I)));
            g_rfft2(s, n, c->gimg, L.rbuf, RS, c->H, nullptr, c->arena_F, I.spec_elems, dst);
        } else {
            Stage st(c, L, "kg_rfft2<image>", n * (Rb(I) + 3 * Cb(I)));
            g_rfft2(s, n, c->gimg, c->arena_img, c->img_stride, c->img_pitch, dst, c->arena_F, I.spec_elems, dst);
        }
        { Stage st(c, L, "kg_abs", n * 2 * Cb(I)); g_abs(s, n, c->arena_F, I.spec_elems, dst, L.gbuf, c->spec_max, I.spec_elems); }
        { Stage st(c, L, "kg_irfft2<image>", n * (Rb(I) + 3 * Cb(I))); g_irfft2(s, n, c->gimg, L.gbuf, c->spec_max, nullptr, L.rbuf + c->r_elems, RS, c->H); }
        { Stage st(c, L, "kg_shift_fix", n * 2 * Rb(I)); g_shift_fix(s, n, L.rbuf + c->r_elems, RS, L.splane, c->s_elems, c->H, c->W); }
    } else {
        if (d_u8) {
            const double N = (double)I.real_elems;
            // (design bytes: the kernel also files the image in the u8 frame store -- N more bytes written, in 16-byte row pieces)
            Stage st(c, L, kname("kA_fwd", c->H / 2, "u8", a_tag(c, I)).c_str(), n * (N + Cb(I)), n * (2 * N + Cb(I)));
            launch_A_fwd_u8(s, n, c->img.g, c->img.t, d_u8, c->img.real_elems, c->W, c->arena_u8, c->u8_stride, c->u8_pitch, dst, L.tmpA, c->spec_max);
        } else {
            Stage st(c, L, kname("kA_fwd", c->H / 2, "plane", a_tag(c, I)).c_str(), n * (Rb(I) + Cb(I)));
            launch_A_fwd_plane(s, n, c->img.g, c->img.t, c->arena_img, c->img_stride, c->img_pitch, dst, L.tmpA, c->spec_max);
        }
        // IFFT(|F|) is real and even and the polar gather only reads the inscribed circle: columns |c| <= Rmax + 1 suffice
        const int need = c->zz_half ? std::min(c->H / 2, c->W / 2) + 1 : 0;
        // columns of the zero-phase image's half-transformed plane that exist at all (the rest is never written nor read)
        const double kept = need > 0 ? (double)shifted_columns(c->img.g, need) / c->W : 1.0;
        // ... and real columns written: each kept column and, for all but column 0 (and W/2), its mirror
        const double wr_cols = need > 0 ? std::min((double)c->W, 2.0 * shifted_columns(c->img.g, need) - 1.0) : (double)c->W;
        { Stage st(c, L, kname("kB", c->W, "fwd_abs_inv", b_tag(c, I)).c_str(), n * 3 * Cb(I), n * (2 + kept) * Cb(I));
          launch_B_fwd_abs_inv(s, n, c->img.g, c->img.t, L.tmpA, c->spec_max, c->arena_F, c->img.spec_elems, dst,
                               L.gbuf, c->spec_max, need); }
        { Stage st(c, L, kname("kA_inv", c->H / 2, "shifted", a_tag(c, I)).c_str(), n * (Cb(I) + Rb(I)), n * (kept * Cb(I) + Rb(I) * wr_cols / c->W));
          launch_A_inv_shifted(s, n, c->img.g, c->img.t, L.gbuf, c->spec_max, L.splane, c->s_elems, need, c->fuse_fix_zero); }
        // RemoveZeroComponent: inside that kernel (mirrored half-plane form), else a launch of its own
        if (!(need > 0 && c->fuse_fix_zero)) launch_fix_zero(s, n, L.splane, c->s_elems, c->H, c->W);
    }
    // ---- polar family: polar(S) and its spectrum (:94)
    if (c->gen_pol) {
        { Stage st(c, L, "kg_polar", n * (Rb(I) + 2 * Rb(P))); g_polar(s, n, L.splane, c->s_elems, c->g_polar_map, L.rbuf, RS, c->H, c->PD, c->PC); }
        Stage st(c, L, "kg_rfft2<polar>", n * (Rb(P) + 3 * Cb(P)));
        g_rfft2(s, n, c->gpol, L.rbuf, RS, c->PD, nullptr, c->arena_P, P.spec_elems, dst);
        return;
    }
    { Stage st(c, L, kname("kA_fwd", c->PD / 2, "polar", a_tag(c, P)).c_str(), n * (Rb(I) + Cb(P)));
      launch_A_fwd_polar(s, n, c->pol.g, c->pol.t, L.splane, c->s_elems, c->H, c->W, c->polar, L.tmpA, c->spec_max); }
    if (defer_polar_B) return;
    { Stage st(c, L, kname("kB", c->PC, "fwd", b_tag(c, P)).c_str(), n * 2 * Cb(P));
      launch_B_fwd(s, n, c->pol.g, c->pol.t, L.tmpA, c->spec_max, c->arena_P, c->pol.spec_elems, dst); }
}

// EstimateTrans (correlation_flow.cc:145-179) for n items.  X spectra: x_fwd ? forward of tmpA lines : arena.
void enqueue_estimate(nik_ctx* c, Lane& L, int n, Family& f, bool x_fwd, const float2* xsrc, size_t x_stride, const int* x_idx,
                      const float2* zsrc, size_t z_stride, const int* z_idx, SurfaceResult* out, int* rot_index, int n_hyp,
                      float2* xstore = nullptr, size_t xstore_stride = 0, const int* xstore_slot = nullptr, Window win = Window()) {
    hipStream_t s = L.stream;
    const double xs_bytes = xstore ? n * Cb(f) : 0.0;        // x_fwd with xstore: the forward spectrum is written out too
    const size_t item_stride = 2 * c->spec_max, plane_stride = c->spec_max;
    if ((&f == &c->pol) ? c->gen_pol : c->gen_img) {
        // any-size family: Kzz and Kxz side by side as 2n planes (x_fwd never set: such contexts do not defer passes)
        const GFamily& gf = (&f == &c->pol) ? c->gpol : c->gimg;
        const char* fam = (&f == &c->pol) ? "polar" : "image";
        auto nm = [&](const char* k) { return std::string(k) + "<" + fam + ">"; };
        if (c->cfg.kernel == 1) launch_energy(s, n, f.g, xsrc, x_stride, x_idx, zsrc, z_stride, z_idx, L.energy);
        { Stage st(c, L, nm("kg_mul").c_str(), n * 4 * Cb(f));
          g_mul(s, n, xsrc, x_stride, x_idx, zsrc, z_stride, z_idx, L.kbuf, item_stride, plane_stride, f.spec_elems, L.maxbuf); }
        { Stage st(c, L, nm("kg_irfft2x2").c_str(), 2 * n * (Rb(f) + 3 * Cb(f)));
          g_irfft2(s, 2 * n, gf, L.kbuf, plane_stride, nullptr, L.rbuf, c->r_elems, f.g.rows); }       // xz = IFFT(xzf)          (:212)
        { Stage st(c, L, nm("kg_kernel").c_str(), 2 * n * 2 * Rb(f));
          g_kernel(s, n, L.rbuf, c->r_elems, f.real_elems, kernel_fn(c), L.energy, L.maxbuf); }        // kernel, max            (:213-214)
        { Stage st(c, L, nm("kg_rfft2x2").c_str(), 2 * n * (Rb(f) + 3 * Cb(f)));
          g_rfft2(s, 2 * n, gf, L.rbuf, c->r_elems, f.g.rows, nullptr, L.kbuf, plane_stride, nullptr); }   // FFT(kernel)          (:215)
        { Stage st(c, L, nm("kg_solve").c_str(), n * 3 * Cb(f));
          g_solve(s, n, L.kbuf, item_stride, plane_stride, L.maxbuf, c->cfg.lambda, L.gbuf, c->spec_max, f.g.cols, f.spec_elems); }   // (:171-172)
        { Stage st(c, L, nm("kg_irfft2").c_str(), n * (Rb(f) + 3 * Cb(f)));
          g_irfft2(s, n, gf, L.gbuf, c->spec_max, nullptr, L.rbuf, 2 * c->r_elems, f.g.rows); }        // g = IFFT(G)            (:173)
        { Stage st(c, L, nm("kg_argmax").c_str(), n * Rb(f));
          g_argmax(s, n, L.rbuf, 2 * c->r_elems, f.g.rows, f.g.cols, L.partials, c->partial_stride, win.row, win.col, win.radius, win.mirror); }
        launch_finalize(s, n, L.partials, c->partial_stride, g_argmax_blocks(f.g.rows, f.g.cols), out, rot_index, n_hyp, c->PD, L.mirror);
        L.mirror = nullptr;
        return;
    }
    if (c->cfg.kernel == 1 && !x_fwd)
        launch_energy(s, n, f.g, xsrc, x_stride, x_idx, zsrc, z_stride, z_idx, L.energy);
    const bool cached = c->kzz_cache;
    if (cached) {
        // key-side kernel Kzz comes from the slot cache (ensure_kzz ran before): only the xz half is computed
        float2* kz = (&f == &c->pol) ? c->arena_KzP : c->arena_KzF;
        unsigned* mz = (&f == &c->pol) ? c->arena_MzP : c->arena_MzF;
        { Stage st(c, L, kname("kB", f.g.cols, x_fwd ? "fwd_mul_inv_x" : "mul_inv_x", b_tag(c, f)).c_str(), n * 3 * Cb(f) + xs_bytes);
          launch_B_mul_inv_x(s, n, f.g, f.t, x_fwd, xsrc, x_stride, x_idx, zsrc, z_stride, z_idx, L.kbuf, item_stride, plane_stride, L.maxbuf,
                             xstore, xstore_stride, xstore_slot); }
        { Stage st(c, L, kname("kA_inv", f.g.rows / 2, "kernel_fwd_x", a_tag(c, f)).c_str(), n * 2 * Cb(f));
          launch_A_inv_kernel_fwd(s, n, f.g, f.t, L.kbuf, item_stride, plane_stride, kernel_fn(c), L.maxbuf, L.energy, 1, 1); }
        { Stage st(c, L, kname("kB", f.g.cols, "solve_cached", b_tag(c, f)).c_str(), n * 3 * Cb(f));
          launch_B_solve_cached(s, n, f.g, f.t, L.kbuf, item_stride, plane_stride, L.maxbuf, kz, f.spec_elems, mz, z_idx,
                                c->cfg.lambda, L.gbuf, c->spec_max); }
    } else {
    // Hermitian-half Kzz: of the zz kernel plane only the columns [0, W/2] (whole tiles) are written, transformed and read
    const double zzk = c->zz_half ? (double)zz_half_columns(f.g) / f.g.cols : 1.0;
    const double zzs = c->zz_half ? (double)(f.g.cols / 2 + 1) / f.g.cols : 1.0;          // (solve_inv reads columns <= W/2 exactly)
    { Stage st(c, L, kname("kB", f.g.cols, x_fwd ? "fwd_mul_inv" : "mul_inv", b_tag(c, f)).c_str(), n * 4 * Cb(f) + xs_bytes, n * (3 + zzk) * Cb(f) + xs_bytes);
      launch_B_mul_inv(s, n, f.g, f.t, x_fwd, xsrc, x_stride, x_idx, zsrc, z_stride, z_idx, L.kbuf, item_stride, plane_stride, L.maxbuf,
                       xstore, xstore_stride, xstore_slot, c->zz_half); }
    { Stage st(c, L, kname("kA_inv", f.g.rows / 2, "kernel_fwd", a_tag(c, f)).c_str(), n * 4 * Cb(f), n * (2 + 2 * zzk) * Cb(f));
      launch_A_inv_kernel_fwd(s, n, f.g, f.t, L.kbuf, item_stride, plane_stride, kernel_fn(c), L.maxbuf, L.energy, 0, 2, c->zz_half); }
    { Stage st(c, L, kname("kB", f.g.cols, "solve_inv", b_tag(c, f)).c_str(), n * 3 * Cb(f), n * (2 + zzs) * Cb(f));
      launch_B_solve_inv(s, n, f.g, f.t, L.kbuf, item_stride, plane_stride, L.maxbuf, c->cfg.lambda, L.gbuf, c->spec_max, c->zz_half); }
    }
    const int nb = argmax_blocks(f.g);
    { Stage st(c, L, kname("kA_inv", f.g.rows / 2, win.row ? "argmax_win" : "argmax", a_tag(c, f)).c_str(), n * Cb(f));
      if (win.row) launch_A_inv_argmax_win(s, n, f.g, f.t, L.gbuf, c->spec_max, L.partials, c->partial_stride, win.row, win.col, win.radius, win.mirror);
      else launch_A_inv_argmax(s, n, f.g, f.t, L.gbuf, c->spec_max, L.partials, c->partial_stride); }
    // (the finalize kernel also writes the results into the call's pinned staging: no device-to-host copy behind it)
    launch_finalize(s, n, L.partials, c->partial_stride, nb, out, rot_index, n_hyp, c->PD, L.mirror);
    L.mirror = nullptr;
}

// ComputePose (correlation_flow.cc:97-143) for n pairs; key/cur slots in the lane's IX_KEY / IX_CUR arrays.
// Leaves raw surface results in the call's h_rot / h_trans (valid after its `done` event).
// polar_in_tmpA: the current frames' polar spectra are still half-transformed in L.tmpA (enqueue_intermedium with
// defer_polar_B): the rotation stage finishes them, stores them in the frame store (slots IX_DST) and uses them.
// img_u8: the current frames' images are read from the u8 frame store (else from the f32 planes).
int enqueue_pose(nik_ctx* c, Lane& L, int n, int not_large_rotation, bool img_u8, bool polar_in_tmpA = false, int win_radius = -1) {
    hipStream_t s = L.stream;
    const int n_hyp = not_large_rotation ? 1 : 2, nt = n * n_hyp;
    Window wrot, wtr;                       // coarse-to-fine: arg-max windows staged in IX_W* (win_radius >= 0)
    if (win_radius >= 0) {
        wrot.row = didx(L, IX_WRR); wrot.col = didx(L, IX_WRC); wrot.radius = win_radius; wrot.mirror = 1;
        wtr.row = didx(L, IX_WTR); wtr.col = didx(L, IX_WTC); wtr.radius = win_radius; wtr.mirror = 0;
    }
    // rotation stage: z = key polar spectrum, x = current polar spectrum
    L.mirror = L.cur->h_rot;
    if (polar_in_tmpA)
        enqueue_estimate(c, L, n, c->pol, true, L.tmpA, c->spec_max, nullptr,
                         c->arena_P, c->pol.spec_elems, didx(L, IX_KEY), L.rot_res, didx(L, IX_ROTIDX), n_hyp,
                         c->arena_P, c->pol.spec_elems, didx(L, IX_DST), wrot);
    else
    enqueue_estimate(c, L, n, c->pol, false, c->arena_P, c->pol.spec_elems, didx(L, IX_CUR),
                     c->arena_P, c->pol.spec_elems, didx(L, IX_KEY), L.rot_res, didx(L, IX_ROTIDX), n_hyp, nullptr, 0, nullptr, wrot);
    // translation items (one per pair and hypothesis); their index arrays were staged by stage_pose_indices()
    // FFT(RotateArray(image, -degree))  (:109 / :116-117): A pass with the rotation gather fused into its load
    if (c->gen_img) {
        // RotateArray + FFT, unfused: the rotated planes, then their spectra in tmpA
        { Stage st(c, L, "kg_rotate", nt * 2 * Rb(c->img));
          g_rotate(s, nt, img_u8 ? c->arena_u8 : nullptr, c->u8_stride, c->u8_pitch, c->arena_img, c->img_stride, c->img_pitch, didx(L, IX_TIMG),
                   c->rot_tab, didx(L, IX_ROTIDX), L.rbuf, 2 * c->r_elems, c->H, c->W); }
        { Stage st(c, L, "kg_rfft2<rotated>", nt * (Rb(c->img) + 3 * Cb(c->img)));
          g_rfft2(s, nt, c->gimg, L.rbuf, 2 * c->r_elems, c->H, nullptr, L.tmpA, c->spec_max, nullptr); }
        L.mirror = L.cur->h_trans;
        enqueue_estimate(c, L, nt, c->img, false, L.tmpA, c->spec_max, nullptr, c->arena_F, c->img.spec_elems,
                         didx(L, IX_TKEY), L.trans_res, nullptr, 1, nullptr, 0, nullptr, wtr);
        return NIK_OK;
    }
    if (img_u8) {
        Stage st(c, L, kname("kA_fwd", c->H / 2, "rot8", a_tag(c, c->img)).c_str(), nt * (1.0 * c->img.real_elems + Cb(c->img)));
        launch_A_fwd_rot8(s, nt, c->img.g, c->img.t, c->arena_u8, c->u8_stride, c->u8_pitch, didx(L, IX_TIMG), c->rot_tab,
                          didx(L, IX_ROTIDX), L.tmpA, c->spec_max);
    } else {
        Stage st(c, L, kname("kA_fwd", c->H / 2, "rot", a_tag(c, c->img)).c_str(), nt * (Rb(c->img) + Cb(c->img)));
        launch_A_fwd_rot(s, nt, c->img.g, c->img.t, c->arena_img, c->img_stride, c->img_pitch, didx(L, IX_TIMG), c->rot_tab,
                         didx(L, IX_ROTIDX), L.tmpA, c->spec_max);
    }
    L.mirror = L.cur->h_trans;
    if (c->cfg.kernel == 1) {
        // gaussian needs sum|X|^2 of the rotated image's spectrum: materialise X (B forward, in place) first
        launch_B_fwd(s, nt, c->img.g, c->img.t, L.tmpA, c->spec_max, L.tmpA, c->spec_max, nullptr);
        enqueue_estimate(c, L, nt, c->img, false, L.tmpA, c->spec_max, nullptr, c->arena_F, c->img.spec_elems,
                         didx(L, IX_TKEY), L.trans_res, nullptr, 1, nullptr, 0, nullptr, wtr);
    } else {
        enqueue_estimate(c, L, nt, c->img, true, L.tmpA, c->spec_max, nullptr, c->arena_F, c->img.spec_elems,
                         didx(L, IX_TKEY), L.trans_res, nullptr, 1, nullptr, 0, nullptr, wtr);
    }
    return NIK_OK;
}

// fill + upload every index array a pose call needs, in ONE host-to-device copy (arrays are contiguous)
int stage_pose_indices(nik_ctx* c, Lane& L, int n, const nik_frame* keys, const nik_frame* curs, int not_large_rotation,
                       bool with_dst) {
    const int n_hyp = not_large_rotation ? 1 : 2, nt = n * n_hyp;
    for (int i = 0; i < n; ++i) { hidx(L, IX_KEY)[i] = keys[i]; hidx(L, IX_CUR)[i] = curs[i]; if (with_dst) hidx(L, IX_DST)[i] = curs[i]; }
    for (int t = 0; t < nt; ++t) {
        const int p = t / n_hyp, hyp = t % n_hyp;
        (void)hyp;
        hidx(L, IX_TIMG)[t] = curs[p];
        hidx(L, IX_TKEY)[t] = keys[p];
    }
    const int used[IX_ROTIDX] = { n, n, with_dst ? n : 0, nt, nt };
    bool same = true;
    for (int w = 0; w < IX_ROTIDX; ++w) same = same && idx_unchanged(L, w, used[w]);
    if (same && !L.idx_force) return NIK_OK;
    HIP_TRY(c, hipMemcpyAsync(L.d_idx, L.cur->h_idx, sizeof(int) * (size_t)L.cap_items * IX_ROTIDX, hipMemcpyHostToDevice, L.stream));
    // (the copy carries whole arrays: whatever lies behind the used prefixes is no longer what the shadow says)
    for (int w = 0; w < IX_ROTIDX; ++w) { L.idx_shadow_n[w] = 0; idx_remember(L, w, used[w]); }
    return NIK_OK;
}

int check_kernel(nik_ctx* c) {
    if (c->cfg.kernel != 0 && c->cfg.kernel != 1) return fail(c, NIK_ERR_INVALID_KERNEL, "Received invalid kernel type");
    return NIK_OK;
}

// contiguous chunk [b, e) of n items for lane li of nl
inline void chunk_of(int n, int nl, int li, int& b, int& e) {
    const int per = (n + nl - 1) / nl;
    b = std::min(n, li * per); e = std::min(n, b + per);
}
// a lane is worth its cross-stream bookkeeping (and the doubled launch count) only with >= 32 items to run: measured,
// batches of 32 are faster on one stream (pyramid workload 16.3 k -> 19.5 k pairs/s)
inline int lanes_for(const nik_ctx* c, int n) { return std::max(1, std::min(c->active_lanes, n / c->lane_items)); }

// Big buffers: plain hipMalloc, or -- tuning switch $NIK_CONTIG=1, round 6 -- physically contiguous memory
// (hipExtMallocWithFlags(hipDeviceMallocContiguous); falls back to hipMalloc when the driver cannot find a contiguous range):
// the experiment behind profiles/r06_placement_contig.txt (do the run-to-run levels of the HBM-bound kernels come from TLB reach?)
// $NIK_VMM=<MiB> (tuning, round 6): the buffer as a reserved address range backed by separately created physical chunks of that
// size (hipMemCreate / hipMemMap) -- the probe of profiles/r06_placement_vmm.txt at library scale.
struct VmmAlloc { size_t size; std::vector<hipMemGenericAllocationHandle_t> handles; };
static std::map<void*, VmmAlloc>& vmm_registry() { static std::map<void*, VmmAlloc> r; return r; }
static hipError_t vmm_malloc(void** p, size_t bytes, size_t chunk_mb, int device) {
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    const size_t chunk = std::max(gran, chunk_mb << 20), n = (bytes + chunk - 1) / chunk;
    void* base = nullptr;
    if ((e = hipMemAddressReserve(&base, n * chunk, 0, nullptr, 0)) != hipSuccess) return e;
    VmmAlloc a; a.size = n * chunk;
    for (size_t k = 0; k < n && e == hipSuccess; ++k) {
        hipMemGenericAllocationHandle_t h;
        if ((e = hipMemCreate(&h, chunk, &prop, 0)) != hipSuccess) break;
        a.handles.push_back(h);
        e = hipMemMap((char*)base + k * chunk, chunk, 0, h, 0);
    }
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (e == hipSuccess) e = hipMemSetAccess(base, a.size, &acc, 1);
    if (e != hipSuccess) { for (auto h : a.handles) (void)hipMemRelease(h); (void)hipMemAddressFree(base, n * chunk); return e; }
    vmm_registry()[base] = std::move(a);
    *p = base;
    return hipSuccess;
}
static hipError_t big_free(void* p) {
    if (!p) return hipSuccess;
    auto it = vmm_registry().find(p);
    if (it == vmm_registry().end()) return hipFree(p);
    (void)hipDeviceSynchronize();
    (void)hipMemUnmap(p, it->second.size);
    for (auto h : it->second.handles) (void)hipMemRelease(h);
    (void)hipMemAddressFree(p, it->second.size);
    vmm_registry().erase(it);
    return hipSuccess;
}
template <class T> hipError_t big_malloc(T** p, size_t bytes) {
    static const bool contig = kcc::tune_env("NIK_CONTIG") && atoi(kcc::tune_env("NIK_CONTIG")) != 0;
    static const int vmm = kcc::tune_env("NIK_VMM") ? atoi(kcc::tune_env("NIK_VMM")) : 0;
    if (vmm > 0) { int dev = 0; (void)hipGetDevice(&dev); return vmm_malloc((void**)p, bytes, (size_t)vmm, dev); }
    if (contig && hipExtMallocWithFlags((void**)p, bytes, hipDeviceMallocContiguous) == hipSuccess) return hipSuccess;
    if (contig) (void)hipGetLastError();
    return hipMalloc((void**)p, bytes);
}

// Spatial partitions ($NIK_LANE_CUS, round 6): "k0,k1,..." gives lane i a stream whose kernels only run on k_i CUs, laid
// behind the CUs of the lanes before it; ONE number k gives every lane the first k CUs.  Mask bit b is CU (b / 8) of XCD b % 8
// (tools/probes/cumask_probe.hip), so any range of bits that starts and ends on a multiple of 8 keeps all eight XCDs with
// equal shares -- which the kernels' blockIdx -> XCD affinity relies on.  Unset: plain streams (the whole chip).
int lane_stream_create(nik_ctx* c, int li, hipStream_t* s) {
    const char* e = kcc::tune_env("NIK_LANE_CUS");
    if (!e || !*e) { HIP_TRY(c, hipStreamCreateWithFlags(s, hipStreamNonBlocking)); return NIK_OK; }
    std::vector<int> k;
    for (const char* p = e; *p;) { k.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
    int first = 0, count = k[0];
    if (k.size() > 1) { for (int i = 0; i < li && i < (int)k.size(); ++i) first += k[i]; count = k[std::min(li, (int)k.size() - 1)]; if (li >= (int)k.size()) first = 0; }
    hipDeviceProp_t prop; HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
    const int ncu = prop.multiProcessorCount;
    if (count <= 0 || first + count > ncu || (first % 8) || (count % 8)) return fail(c, NIK_ERR_INVALID_ARG, "NIK_LANE_CUS: lane %d would own CUs [%d, %d) of %d (multiples of 8 only)", li, first, first + count, ncu);
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int b = first; b < first + count; ++b) mask[b >> 5] |= 1u << (b & 31);
    HIP_TRY(c, hipExtStreamCreateWithCUMask(s, (uint32_t)mask.size(), mask.data()));
    return NIK_OK;
}

int lane_alloc(nik_ctx* c, Lane& L, int nl, int li) {
    L.cap_items = c->max_items;
    L.seen.assign(nl, 0); L.seen_tail.assign(nl, 0);
    { const int rc = lane_stream_create(c, li, &L.stream); if (rc) return rc; }
    HIP_TRY(c, hipEventCreateWithFlags(&L.write_ev, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&L.tail_ev, hipEventDisableTiming));
    if (const char* sl = kcc::tune_env("NIK_LANE_SLAB")) {
        // (round 6 experiment: the three spectrum work buffers of a lane as ONE allocation, pieces `pad` MiB apart beyond their
        // size -- does the placement lottery of profiles/r06_placement_probe.txt change when their relative offsets are ours?)
        const size_t unit = sizeof(float2) * c->spec_max * c->max_items, pad = (size_t)std::max(0, atoi(sl)) << 20;
        const size_t a = (unit + (2u << 20) - 1) / (2u << 20) * (2u << 20) + pad;
        char* slab = nullptr;
        HIP_TRY(c, hipMalloc(&slab, 4 * a + pad));
        L.tmpA = (float2*)slab; L.kbuf = (float2*)(slab + a); L.gbuf = (float2*)(slab + 3 * a + (pad ? pad / 2 : 0)); L.slab = slab;
    } else {
    HIP_TRY(c, big_malloc(&L.tmpA, sizeof(float2) * c->spec_max * c->max_items));
    HIP_TRY(c, big_malloc(&L.kbuf, sizeof(float2) * c->spec_max * 2 * c->max_items));
    HIP_TRY(c, big_malloc(&L.gbuf, sizeof(float2) * c->spec_max * c->max_items));
    }
    // (+16: the polar gather stages whole 16-float chunks, the last of which may start at the plane's last pixel)
    HIP_TRY(c, big_malloc(&L.splane, sizeof(float) * (c->s_elems * c->max_batch + 16)));
    HIP_TRY(c, hipMemset(L.splane, 0, sizeof(float) * (c->s_elems * c->max_batch + 16)));      // zero borders are never overwritten
    if (c->generic) HIP_TRY(c, hipMalloc(&L.rbuf, sizeof(float) * 2 * c->r_elems * c->max_items));
    HIP_TRY(c, hipMalloc(&L.partials, sizeof(Partial) * c->partial_stride * c->max_items));
    HIP_TRY(c, hipMalloc(&L.maxbuf, sizeof(unsigned) * 2 * KCC_MAXPARTS * c->max_items));
    HIP_TRY(c, hipMemset(L.maxbuf, 0, sizeof(unsigned) * 2 * KCC_MAXPARTS * c->max_items));
    HIP_TRY(c, hipMalloc(&L.energy, sizeof(float) * 2 * c->max_items));
    HIP_TRY(c, hipMemset(L.energy, 0, sizeof(float) * 2 * c->max_items));
    HIP_TRY(c, hipMalloc(&L.rot_res, sizeof(SurfaceResult) * c->max_batch));
    HIP_TRY(c, hipMalloc(&L.trans_res, sizeof(SurfaceResult) * c->max_items));
    HIP_TRY(c, hipMalloc(&L.d_idx, sizeof(int) * c->max_items * IX_COUNT));
    L.idx_shadow.assign((size_t)c->max_items * IX_ROTIDX, -1);
    for (Call& call : L.ring) {
        HIP_TRY(c, hipHostMalloc(&call.h_idx, sizeof(int) * c->max_items * IX_COUNT));
        HIP_TRY(c, hipHostMalloc(&call.h_rot, sizeof(SurfaceResult) * c->max_batch));
        HIP_TRY(c, hipHostMalloc(&call.h_trans, sizeof(SurfaceResult) * c->max_items));
        HIP_TRY(c, hipEventCreateWithFlags(&call.done, hipEventDisableTiming));
    }
    return NIK_OK;
}
void lane_free(Lane& L) {
    if (L.stream) (void)hipStreamSynchronize(L.stream);
    if (L.slab) (void)hipFree(L.slab); else { (void)big_free(L.tmpA); (void)big_free(L.kbuf); (void)big_free(L.gbuf); }
    (void)big_free(L.splane); (void)hipFree(L.u8tmp); (void)hipFree(L.rbuf); (void)hipFree(L.partials);
    (void)hipFree(L.maxbuf); (void)hipFree(L.energy); (void)hipFree(L.rot_res); (void)hipFree(L.trans_res); (void)hipFree(L.d_idx);
    for (Call& call : L.ring) {
        if (call.h_idx) (void)hipHostFree(call.h_idx);
        if (call.h_rot) (void)hipHostFree(call.h_rot);
        if (call.h_trans) (void)hipHostFree(call.h_trans);
        if (call.done) (void)hipEventDestroy(call.done);
    }
    for (auto& g : L.graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (L.write_ev) (void)hipEventDestroy(L.write_ev);
    if (L.tail_ev) (void)hipEventDestroy(L.tail_ev);
    if (L.stream) (void)hipStreamDestroy(L.stream);
}

// streams and work buffers of lanes [0, n): created on first use
int ensure_lanes(nik_ctx* c, int n) {
    for (int li = 0; li < n && li < (int)c->lanes.size(); ++li)
        if (!c->lanes[li].stream) {
            int rc = lane_alloc(c, c->lanes[li], (int)c->lanes.size(), li);
            if (rc) return rc;
            if (c->up_fenced >= 0) HIP_TRY(c, hipStreamWaitEvent(c->lanes[li].stream, c->up_ev[c->up_fenced & 3], 0));   // nik_upload_fence
        }
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
    // The tiled kernels are compile-time plans: half-rows / line lengths of a closed set, 16-column tiles (W, PC multiples of
    // 16) and an aspect ratio within 2:1 (every de-rotated source coordinate then stays within one period: single-step
    // BORDER_WRAP).  Every other geometry the reference accepts (correlation_flow.cc:53-77: any size with even rows) runs the
    // any-size family (kcc_generic.hip): slower, same results.  $NIK_GENERIC=1 forces it (tests compare the two families).
    // The choice is per plane family: a 640 x 480 camera with a 720 x 64 polar plane keeps the tiled image kernels.
    const int force = getenv("NIK_GENERIC") ? atoi(getenv("NIK_GENERIC")) : 0;      // 1 = both families, 2 = polar family only, 4 = image family only (tests)
    const bool gen_img = !(fft_half_supported(H / 2) && fft_line_supported(W) && W % 16 == 0 && !(H / 2 + 2 > W || W / 2 + 2 > H)) || force == 1 || (force & 4);
    bool gen_pol = !(fft_half_supported(PD / 2) && fft_line_supported(PC) && PC % 16 == 0) || force == 1 || (force & 2);
    const bool generic = gen_img || gen_pol;
    if (generic && (std::max({ H, W, PD, PC }) > 8192 || H < 4 || W < 4 || PD < 4 || PC < 4))
        return fail(nullptr, NIK_ERR_UNSUPPORTED_SIZE, "any-size path: lengths from 4 to 8192 (got %dx%d / polar %dx%d)", H, W, PD, PC);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, NIK_ERR_HIP, "no HIP device available (the HIP path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, NIK_ERR_INVALID_ARG, "device %d out of range", device);
    nik_ctx* c = new nik_ctx();
    c->cfg = *cfg; c->cfg.height = H; c->cfg.width = W;      // correlation_flow.cc:40-41
    c->H = H; c->W = W; c->PD = PD; c->PC = PC; c->max_batch = max_batch; c->max_frames = max_frames; c->device = device;
    c->max_items = 2 * max_batch;
    c->generic = generic; c->gen_img = gen_img; c->gen_pol = gen_pol;
    auto bail = [&](int rc) { g_create_error = c->err; nik_destroy(c); return rc; };
#define TRY_C(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fail(c, NIK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); return bail(NIK_ERR_HIP); } } while (0)
    TRY_C(hipSetDevice(device));
    int rc;
    if ((rc = family_init(c, c->img, H, W)) || (rc = family_init(c, c->pol, PD, PC))) return bail(rc);
    c->spec_max = std::max(c->img.spec_elems, c->pol.spec_elems);
    c->s_elems = ((size_t)(W + 1) * (H + 2) + 15) / 16 * 16;      // (items 64-byte aligned: the aligned polar staging relies on it)
    c->r_elems = std::max(c->img.real_elems, c->pol.real_elems);
    // the tiled polar gather stages annulus segments of THIS image geometry in LDS: an image size whose segments do not fit
    // sends the polar family to the any-size kernels as well
    // (only THAT outcome -- NIK_ERR_UNSUPPORTED_SIZE -- falls back; an allocation or copy error of the table upload is an error)
    if (!c->gen_pol) {
        const int prc = build_polar_table(c);
        if (prc == NIK_ERR_UNSUPPORTED_SIZE) { c->gen_pol = true; c->generic = true; c->err.clear(); }
        else if (prc) return bail(prc);
    }
    if (c->generic && (rc = generic_init(c))) return bail(rc);
    c->partial_stride = std::max(c->gen_img ? g_argmax_blocks(H, W) : argmax_blocks(c->img.g), c->gen_pol ? g_argmax_blocks(PD, PC) : argmax_blocks(c->pol.g));
    // column pitch: >= H + 4 (wrap rows), a multiple of 32 floats (columns start on 128-byte lines) and an ODD multiple
    // (no power-of-two stride across HBM channels)
    c->img_pitch = ((H + 4 + 31) / 32) * 32;
    if (((c->img_pitch / 32) & 1) == 0) c->img_pitch += 32;
    c->img_stride = (size_t)W * c->img_pitch;
    TRY_C(hipMalloc(&c->arena_img, sizeof(float) * c->img_stride * max_frames));
    c->u8_pitch = W + 16; c->u8_stride = (size_t)c->u8_pitch * H;
    TRY_C(big_malloc(&c->arena_u8, c->u8_stride * max_frames));
    TRY_C(big_malloc(&c->arena_F, sizeof(float2) * c->img.spec_elems * max_frames));
    TRY_C(big_malloc(&c->arena_P, sizeof(float2) * c->pol.spec_elems * max_frames));
    TRY_C(hipMalloc(&c->arena_KzF, sizeof(float2) * c->img.spec_elems * max_frames));
    TRY_C(hipMalloc(&c->arena_KzP, sizeof(float2) * c->pol.spec_elems * max_frames));
    TRY_C(hipMalloc(&c->arena_MzF, sizeof(unsigned) * max_frames));
    TRY_C(hipMalloc(&c->arena_MzP, sizeof(unsigned) * max_frames));
    c->slot_kzz.assign(max_frames, 0);
    if (const char* e = getenv("NIK_KZZ_CACHE")) c->kzz_cache = atoi(e) != 0;
    if (const char* e = kcc::tune_env("NIK_FUSE_POLAR")) c->fuse_polar = atoi(e) != 0;
    if (const char* e = kcc::tune_env("NIK_ZZ_HALF")) c->zz_half = atoi(e) != 0;
    if (const char* e = kcc::tune_env("NIK_ALT_ORDER")) c->alt_order = atoi(e) != 0;
    if (const char* e = kcc::tune_env("NIK_FUSE_FIX_ZERO")) c->fuse_fix_zero = atoi(e) != 0;
    if (const char* e = kcc::tune_env("NIK_CHUNK")) c->chunk_pairs = std::max(0, atoi(e));
    if (const char* e = kcc::tune_env("NIK_LANE_ITEMS")) c->lane_items = std::max(1, atoi(e));
    if (const char* e = getenv("NIK_GRAPH")) c->graph_max = std::max(0, atoi(e));
    c->slot_kind.assign(max_frames, 0);
    c->slot_ready.assign(max_frames, 0); c->slot_lane.assign(max_frames, -1); c->slot_seq.assign(max_frames, 0); c->slot_rd.assign((size_t)max_frames * 4, 0);
    int nl = 3;                                   // measured: 3 streams beat 2 by 1-3 % at 256 pairs per call (tools/sweep_chunk.sh)
    if (const char* e = getenv("NIK_STREAMS")) nl = atoi(e);
    // (a call of n items uses n / 32 streams at most -- lanes_for(): a context whose batches are small never needs the others;
    // unused streams are not free: the pyramid workload lost 30 % to two idle lanes per level)
    nl = std::max(1, std::min({ 4, nl, std::max(1, max_batch / 32) }));
    // lane 0 now, the others when a call first needs them (ensure_lanes): a stream that exists takes a hardware queue slot
    // from the ones that work, and a context that is switched to one stream right after creation (the pyramid's levels)
    // should never pay for the others
    c->lanes.resize(nl); c->active_lanes = nl;
    if ((rc = lane_alloc(c, c->lanes[0], nl, 0))) return bail(rc);
    TRY_C(hipMalloc(&c->d_u8, (size_t)H * W));
    TRY_C(hipMalloc(&c->d_scratch, sizeof(float) * std::max(c->img.real_elems, 2 * c->spec_max) * 2));
    if (c->generic) { c->fuse_polar = false; c->kzz_cache = false; c->graph_max = 0; }    // (the any-size family has no fused / cached / captured forms)
    if ((rc = build_rot_table(c))) return bail(rc);
    TRY_C(hipDeviceSynchronize());
#undef TRY_C
    *out = c;
    return NIK_OK;
}

void nik_destroy(nik_ctx* c) {
    if (!c) return;
    for (Lane& L : c->lanes) lane_free(L);
    for (Family* f : { &c->img, &c->pol }) for (float2* p : f->d_tw) (void)hipFree(p);
    (void)big_free(c->arena_u8); (void)hipFree(c->arena_img); (void)big_free(c->arena_F); (void)big_free(c->arena_P);
    (void)hipFree(c->arena_KzF); (void)hipFree(c->arena_KzP); (void)hipFree(c->arena_MzF); (void)hipFree(c->arena_MzP);
    (void)hipFree(c->ud_map1); (void)hipFree(c->ud_map2); (void)hipFree(c->d_stats);
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    if (c->stats_stream) { (void)hipStreamSynchronize(c->stats_stream); (void)hipStreamDestroy(c->stats_stream); }
    if (c->stats_done) (void)hipEventDestroy(c->stats_done);
    if (c->up_stream) { (void)hipStreamSynchronize(c->up_stream); (void)hipStreamDestroy(c->up_stream); }
    for (hipEvent_t e : c->up_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->up_pin_ev) if (e) (void)hipEventDestroy(e);
    if (c->up_after_ev) (void)hipEventDestroy(c->up_after_ev);
    for (uint8_t* q : c->up_pin) if (q) (void)hipHostFree(q);
    for (float2* q : c->g_tw) (void)hipFree(q);
    (void)hipFree(c->g_polar_map);
    (void)hipFree(c->d_u8); (void)hipFree(c->d_scratch); (void)hipFree(const_cast<uint32_t*>(c->polar.chunks)); (void)hipFree(const_cast<int*>(c->polar.seg_first));
    (void)hipFree(const_cast<uint4*>(c->polar.pts)); (void)hipFree(c->rot_tab); (void)hipFree(c->rot_one);
    for (hipEvent_t e : c->chain_ev) if (e) (void)hipEventDestroy(e);
    if (c->fence_ev) (void)hipEventDestroy(c->fence_ev);
    for (auto& r : c->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : c->prof_pool) (void)hipEventDestroy(e);
    delete c;
}

const char* nik_last_error(const nik_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int nik_get_dims(const nik_ctx* c, int dims[6]) {
    if (!c || !dims) return NIK_ERR_INVALID_ARG;
    dims[0] = c->H; dims[1] = c->W; dims[2] = c->PD; dims[3] = c->PC; dims[4] = c->max_batch; dims[5] = c->max_frames;
    return NIK_OK;
}
int nik_device(const nik_ctx* c) { return c ? c->device : -1; }
int nik_is_generic(const nik_ctx* c) { return c ? ((c->gen_img ? 1 : 0) | (c->gen_pol ? 2 : 0)) : NIK_ERR_INVALID_ARG; }
void* nik_stream(const nik_ctx* c) { return c ? (void*)c->lanes[0].stream : nullptr; }

int nik_set_streams(nik_ctx* c, int n) {
    if (!c) return NIK_ERR_INVALID_ARG;
    int rc = drain_all(c);
    if (rc) return rc;
    c->active_lanes = std::max(1, std::min((int)c->lanes.size(), n));
    return c->active_lanes;
}

int nik_set_chunk(nik_ctx* c, int pairs) {
    if (!c || pairs < 0) return NIK_ERR_INVALID_ARG;
    const int old = c->chunk_pairs;
    c->chunk_pairs = pairs;
    return old;
}

int nik_set_residual_stats(nik_ctx* c, int enable) {
    if (!c) return NIK_ERR_INVALID_ARG;
    int rc = drain_all(c);
    if (rc) return rc;
    if (enable && !c->d_stats) {
        HIP_TRY(c, hipMalloc(&c->d_stats, sizeof(double) * 4 * (KCC_STATS_PARTS + 1)));           // chunk partials + the total
        HIP_TRY(c, hipMemset(c->d_stats, 0, sizeof(double) * 4 * (KCC_STATS_PARTS + 1)));
        HIP_TRY(c, hipHostMalloc(&c->h_stats, sizeof(double) * 4));
        HIP_TRY(c, hipStreamCreateWithFlags(&c->stats_stream, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->stats_done, hipEventDisableTiming));
    }
    c->want_stats = enable != 0; c->stats_lanes = 0; c->stats_parts = 0;
    return NIK_OK;
}

// total of the latest batch call's per-lane partials, enqueued on the statistics stream behind every lane's share
int nik_residual_stats_dev(nik_ctx* c, double** d_total, void** stream) {
    if (!c || !d_total) return NIK_ERR_INVALID_ARG;
    if (!c->want_stats) return fail(c, NIK_ERR_NOT_READY, "residual statistics are off (nik_set_residual_stats)");
    for (int li = 0; li < c->stats_lanes; ++li) HIP_TRY(c, hipStreamWaitEvent(c->stats_stream, c->lanes[li].tail_ev, 0));
    launch_stats_sum(c->stats_stream, c->d_stats, c->stats_parts, c->d_stats + 4 * KCC_STATS_PARTS);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->stats_done, c->stats_stream));
    c->stats_pending = true;
    *d_total = c->d_stats + 4 * KCC_STATS_PARTS;
    if (stream) *stream = (void*)c->stats_stream;
    return NIK_OK;
}

int nik_residual_stats(nik_ctx* c, double out[4]) {
    if (!c || !out) return NIK_ERR_INVALID_ARG;
    double* d = nullptr;
    int rc = nik_residual_stats_dev(c, &d, nullptr);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->h_stats, d, sizeof(double) * 4, hipMemcpyDeviceToHost, c->stats_stream));
    HIP_TRY(c, hipStreamSynchronize(c->stats_stream));
    memcpy(out, c->h_stats, sizeof(double) * 4);
    return NIK_OK;
}

// batches of <= max_pairs stored frames (nik_pose_batch / nik_pose / nik_match chunks on one stream, Kzz cache off) replay a
// captured hipGraph instead of ~16 separate launches; 0 switches it off
int nik_set_graphs(nik_ctx* c, int max_pairs) {
    if (!c) return NIK_ERR_INVALID_ARG;
    int rc = drain_all(c);
    if (rc) return rc;
    c->graph_max = c->generic ? 0 : std::max(0, max_pairs);
    return NIK_OK;
}

int nik_set_kzz_cache(nik_ctx* c, int enable) {
    if (!c) return NIK_ERR_INVALID_ARG;
    int rc = drain_all(c);
    if (rc) return rc;
    c->kzz_cache = enable != 0 && !c->generic;                // (the any-size family recomputes Kzz: same results)
    return NIK_OK;
}

int nik_set_undistort(nik_ctx* c, const int16_t* map1, const uint16_t* map2) {
    if (!c || ((map1 == nullptr) != (map2 == nullptr))) return fail(c, NIK_ERR_INVALID_ARG, "both maps or neither");
    int rc = drain_all(c);
    if (rc) return rc;
    for (Lane& L : c->lanes) if (L.stream) HIP_TRY(c, hipStreamSynchronize(L.stream));
    (void)hipFree(c->ud_map1); (void)hipFree(c->ud_map2); c->ud_map1 = nullptr; c->ud_map2 = nullptr;
    if (!map1) return NIK_OK;
    const size_t n = c->img.real_elems;
    int16_t* m1 = nullptr; uint16_t* m2 = nullptr;             // installed only when both copies succeeded
    hipError_t e = hipMalloc(&m1, n * 2 * sizeof(int16_t));
    if (e == hipSuccess) e = hipMalloc(&m2, n * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemcpy(m1, map1, n * 2 * sizeof(int16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m2, map2, n * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(m1); (void)hipFree(m2); return fail(c, NIK_ERR_HIP, "undistortion maps: %s", hipGetErrorString(e)); }
    for (Lane& L : c->lanes)                                  // undistorted frames of one call (input of the first FFT pass)
        if (!L.u8tmp && e == hipSuccess) e = hipMalloc(&L.u8tmp, n * (size_t)c->max_batch);
    if (e != hipSuccess) { (void)hipFree(m1); (void)hipFree(m2); return fail(c, NIK_ERR_HIP, "undistortion staging: %s", hipGetErrorString(e)); }
    c->ud_map1 = m1; c->ud_map2 = m2;
    return NIK_OK;
}

int nik_undistort_dev(nik_ctx* c, int n, const uint8_t* d_in, uint8_t* d_out) {
    if (!c || !d_in || !d_out || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    if (!c->ud_map1) return fail(c, NIK_ERR_INVALID_ARG, "no undistortion maps installed (nik_set_undistort)");
    if (n == 0) return NIK_OK;
    int rc = drain_all(c);
    if (rc) return rc;
    launch_undistort_u8(c->lanes[0].stream, n, d_in, d_out, c->ud_map1, c->ud_map2, c->H, c->W);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->lanes[0].stream));
    return NIK_OK;
}

int nik_synchronize(nik_ctx* c) {
    if (!c) return NIK_ERR_INVALID_ARG;
    int rc = drain_all(c);
    if (rc) return rc;
    for (int li = 0; li < c->active_lanes; ++li) if (c->lanes[li].stream) HIP_TRY(c, hipStreamSynchronize(c->lanes[li].stream));
    return NIK_OK;
}

int nik_intermedium_batch_dev(nik_ctx* c, int n, const uint8_t* d_gray, const nik_frame* dst) {
    if (!c || !d_gray || !dst || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    if (n == 0) return NIK_OK;
    if (n > c->max_batch) return fail(c, NIK_ERR_CAPACITY, "batch %d exceeds max_batch %d", n, c->max_batch);
    int rc;
    for (int i = 0; i < n; ++i) if ((rc = check_slot(c, dst[i], false))) return rc;
    const int nl = lanes_for(c, n);
    const bool rot = c->lane_rot && c->active_lanes > 1;
    const int LN = rot ? c->active_lanes : nl, base = rot ? c->lane_base : 0;
    if ((rc = ensure_lanes(c, LN))) return rc;
    if (rot) c->lane_base = (base + nl) % LN;
    for (int ci = 0; ci < nl; ++ci) {
        int b, e; chunk_of(n, nl, ci, b, e);
        const int m = e - b; if (m <= 0) continue;
        const int li = (base + ci) % LN;
        Lane& L = c->lanes[li];
        if ((rc = begin_call(c, L))) return rc;
        for (int i = 0; i < m; ++i) { if ((rc = depend_for_write(c, L, li, dst[b + i]))) return rc; hidx(L, IX_DST)[i] = dst[b + i]; }
        if ((rc = upload_idx(c, L, IX_DST, m))) return rc;
        enqueue_intermedium(c, L, m, d_gray + (size_t)b * c->img.real_elems);
        HIP_TRY(c, hipGetLastError());
        if ((rc = mark_written(c, L, li, dst + b, m, 1)) || (rc = end_call(c, L))) return rc;
    }
    return NIK_OK;
}

int nik_intermedium_u8(nik_ctx* c, const uint8_t* gray, int stride, nik_frame dst) {
    if (!c || !gray) return fail(c, NIK_ERR_INVALID_ARG, "null argument");
    if (stride < c->W) return fail(c, NIK_ERR_INVALID_ARG, "stride %d smaller than width %d", stride, c->W);
    int rc;
    if ((rc = drain_all(c))) return rc;                      // d_u8 staging is shared: one host image at a time
    HIP_TRY(c, hipMemcpy2DAsync(c->d_u8, c->W, gray, stride, c->W, c->H, hipMemcpyHostToDevice, c->lanes[0].stream));
    const int keep = c->active_lanes; c->active_lanes = 1;
    rc = nik_intermedium_batch_dev(c, 1, c->d_u8, &dst);
    c->active_lanes = keep;
    if (rc) return rc;
    return drain_all(c);
}

// ---- host frames -> device, on the upload stream ---------------------------------------------------------------------
// The reference's caller hands over host images one by one (main.cpp:55-65, map_builder.cc:30-33).  A streamed caller uploads the
// next window of frames while the current one is registered: the copies run on the context's own stream, a pinned source
// (hipHostMalloc / hipHostRegister: what a camera driver's DMA ring is) is read by the copy engine directly, a pageable one goes
// through two pinned staging buffers (the CPU copy of piece k+1 overlaps the DMA of piece k).  Returns a ticket; the compute
// lanes wait for it on the device once nik_upload_fence(ticket) has been called (NOT before: an upload enqueued behind a
// fence does not delay the work that fence covers -- that is what lets window k+1 travel while window k computes).
int nik_upload_u8_async(nik_ctx* c, int n, const uint8_t* gray, int stride, size_t frame_stride, uint8_t* d_dst) {
    if (!c || !gray || !d_dst || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    if (stride < c->W || frame_stride < (size_t)stride * (size_t)c->H) return fail(c, NIK_ERR_INVALID_ARG, "stride %d / frame stride %zu too small for %d x %d frames", stride, frame_stride, c->H, c->W);
    if (!c->up_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
        for (hipEvent_t& e : c->up_ev) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t& e : c->up_pin_ev) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const size_t fb = (size_t)c->H * c->W;
    hipPointerAttribute_t at{};
    // pinned host memory -- and anything else the copy engine can read by itself (device, managed) -- is copied directly; only
    // plain pageable memory is staged
    const bool known = hipPointerGetAttributes(&at, gray) == hipSuccess;
    (void)hipGetLastError();                                  // (an unregistered pointer reports an error: not ours)
    const bool pinned = known && (at.type == hipMemoryTypeHost || at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged);
    if (pinned) {
        if (frame_stride == (size_t)stride * (size_t)c->H)   // the frames form one tall image: a single 2-D copy
            HIP_TRY(c, hipMemcpy2DAsync(d_dst, c->W, gray, stride, c->W, (size_t)c->H * n, hipMemcpyDefault, c->up_stream));
        else
            for (int i = 0; i < n; ++i)
                HIP_TRY(c, hipMemcpy2DAsync(d_dst + (size_t)i * fb, c->W, gray + (size_t)i * frame_stride, stride, c->W, c->H, hipMemcpyDefault, c->up_stream));
    } else {
        const int per = 8;                                    // frames per staging piece
        if (c->up_pin_bytes < fb * per) {
            for (int k = 0; k < 2; ++k) { if (c->up_pin_busy[k]) { HIP_TRY(c, hipEventSynchronize(c->up_pin_ev[k])); c->up_pin_busy[k] = false; } if (c->up_pin[k]) (void)hipHostFree(c->up_pin[k]); c->up_pin[k] = nullptr; }
            for (int k = 0; k < 2; ++k) HIP_TRY(c, hipHostMalloc(&c->up_pin[k], fb * per));
            c->up_pin_bytes = fb * per;
        }
        for (int b = 0; b < n; b += per) {
            const int m = std::min(per, n - b), k = c->up_pin_next; c->up_pin_next ^= 1;
            if (c->up_pin_busy[k]) { HIP_TRY(c, hipEventSynchronize(c->up_pin_ev[k])); c->up_pin_busy[k] = false; }
            for (int i = 0; i < m; ++i) {
                const uint8_t* src = gray + (size_t)(b + i) * frame_stride;
                uint8_t* dst = c->up_pin[k] + (size_t)i * fb;
                if (stride == c->W) memcpy(dst, src, fb);
                else for (int r = 0; r < c->H; ++r) memcpy(dst + (size_t)r * c->W, src + (size_t)r * stride, c->W);
            }
            HIP_TRY(c, hipMemcpyAsync(d_dst + (size_t)b * fb, c->up_pin[k], fb * m, hipMemcpyHostToDevice, c->up_stream));
            HIP_TRY(c, hipEventRecord(c->up_pin_ev[k], c->up_stream));
            c->up_pin_busy[k] = true;
        }
    }
    const unsigned t = c->up_seq++;
    HIP_TRY(c, hipEventRecord(c->up_ev[t & 3], c->up_stream));
    return (int)(t & 0x3FFFFFFF);
}
// every compute lane waits (on the device) for the upload with this ticket; at most four uploads may be outstanding
int nik_upload_fence(nik_ctx* c, int ticket) {
    if (!c || ticket < 0 || !c->up_stream) return fail(c, NIK_ERR_INVALID_ARG, "no such upload");
    if ((unsigned)ticket + 4 < (c->up_seq & 0x3FFFFFFFu) ) return fail(c, NIK_ERR_INVALID_ARG, "upload ticket %d is older than the four tracked uploads", ticket);
    if ((unsigned)ticket >= (c->up_seq & 0x3FFFFFFFu)) return fail(c, NIK_ERR_INVALID_ARG, "upload ticket %d was never issued", ticket);   // (its event is unrecorded or stale)
    // (only the lanes that exist: a stream that is merely created takes a hardware queue from the ones that work; a lane created
    // later waits for the latest fenced upload when it is created -- ensure_lanes)
    for (int li = 0; li < c->active_lanes; ++li)
        if (c->lanes[li].stream) HIP_TRY(c, hipStreamWaitEvent(c->lanes[li].stream, c->up_ev[ticket & 3], 0));
    c->up_fenced = ticket;
    return NIK_OK;
}
// device memory on the context's GPU for callers that do not link the HIP runtime themselves (the host-side tracker's upload ring)
int nik_dev_malloc(nik_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return fail(c, NIK_ERR_INVALID_ARG, "null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMalloc(out, bytes ? bytes : 1));
    return NIK_OK;
}
int nik_dev_free(nik_ctx* c, void* p) {
    if (!c) return NIK_ERR_INVALID_ARG;
    if (p) { HIP_TRY(c, hipSetDevice(c->device)); HIP_TRY(c, hipFree(p)); }
    return NIK_OK;
}
// the upload stream waits (on the device) for everything enqueued on the compute lanes so far: an upload enqueued next may
// overwrite a device buffer those calls still read (the tracker's upload ring reusing a window buffer)
int nik_upload_after_compute(nik_ctx* c) {
    if (!c) return NIK_ERR_INVALID_ARG;
    if (!c->up_stream) return NIK_OK;                       // (nothing uploaded yet: the first upload creates the stream)
    HIP_TRY(c, hipSetDevice(c->device));
    for (int li = 0; li < c->active_lanes; ++li) {
        Lane& L = c->lanes[li];
        if (!L.stream) continue;
        if (!c->up_after_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->up_after_ev, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->up_after_ev, L.stream));
        HIP_TRY(c, hipStreamWaitEvent(c->up_stream, c->up_after_ev, 0));
    }
    return NIK_OK;
}
// host-side wait: the SOURCE buffers of every upload enqueued so far may be reused
int nik_upload_wait(nik_ctx* c) {
    if (!c) return NIK_ERR_INVALID_ARG;
    if (c->up_stream) HIP_TRY(c, hipStreamSynchronize(c->up_stream));
    return NIK_OK;
}

int nik_intermedium_f32(nik_ctx* c, const float* image, nik_frame dst) {
    if (!c || !image) return fail(c, NIK_ERR_INVALID_ARG, "null argument");
    int rc;
    if ((rc = drain_all(c)) || (rc = check_slot(c, dst, false))) return rc;
    Lane& L = c->lanes[0];
    if ((rc = begin_call(c, L)) || (rc = depend_for_write(c, L, 0, dst))) return rc;
    HIP_TRY(c, hipMemcpy2DAsync(c->arena_img + (size_t)dst * c->img_stride, sizeof(float) * c->img_pitch, image, sizeof(float) * c->H,
                                sizeof(float) * c->H, c->W, hipMemcpyHostToDevice, L.stream));
    if (!c->gen_img) launch_img_wrap(L.stream, c->arena_img + (size_t)dst * c->img_stride, c->H, c->W, c->img_pitch);   // (the any-size rotate wraps by index)
    hidx(L, IX_DST)[0] = dst;
    if ((rc = upload_idx(c, L, IX_DST, 1))) return rc;
    enqueue_intermedium(c, L, 1, nullptr);
    HIP_TRY(c, hipGetLastError());
    if ((rc = mark_written(c, L, 0, &dst, 1, 2)) || (rc = end_call(c, L))) return rc;
    return drain_all(c);
}

int nik_frame_export(nik_ctx* c, nik_frame f, float* image, float* fft_result, float* fft_polar) {
    if (!c) return NIK_ERR_INVALID_ARG;
    int rc;
    if ((rc = nik_synchronize(c)) || (rc = check_slot(c, f, true))) return rc;
    hipStream_t s = c->lanes[0].stream;
    if (image && !(c->slot_kind[f] & 2)) {                  // the frame arrived as u8: hand out ConvertMatToNormalizedArray of it
        Lane& L = c->lanes[0];
        if ((rc = begin_call(c, L)) || (rc = ensure_f32_images(c, L, 0, 1, &f)) || (rc = end_call(c, L)) || (rc = drain_all(c))) return rc;
    }
    if (image) HIP_TRY(c, hipMemcpy2DAsync(image, sizeof(float) * c->H, c->arena_img + (size_t)f * c->img_stride, sizeof(float) * c->img_pitch,
                                           sizeof(float) * c->H, c->W, hipMemcpyDeviceToHost, s));
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
    if ((rc = nik_synchronize(c)) || (rc = check_slot(c, f, false))) return rc;
    hipStream_t s = c->lanes[0].stream;
    float2* scratch = reinterpret_cast<float2*>(c->d_scratch);
    if (image) {
        HIP_TRY(c, hipMemcpy2DAsync(c->arena_img + (size_t)f * c->img_stride, sizeof(float) * c->img_pitch, image, sizeof(float) * c->H,
                                    sizeof(float) * c->H, c->W, hipMemcpyHostToDevice, s));
        if (!c->gen_img) launch_img_wrap(s, c->arena_img + (size_t)f * c->img_stride, c->H, c->W, c->img_pitch);
        c->slot_ready[f] |= 1; c->slot_kind[f] = 2;
    }
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
    c->slot_lane[f] = -1;                                   // host-synchronous write: visible to every lane
    c->slot_kzz[f] = 0;
    if (fft_result && fft_polar) c->slot_ready[f] |= 2;
    return NIK_OK;
}

// Build the Kzz cache of every listed key slot that lacks it (both families), on lane 0:
//   Kzz' = FFT( kernel( IFFT(|Z|^2) ) ) (not yet divided by its max) and Mzz = max|kernel|   (correlation_flow.cc:160,164)
static int ensure_kzz_run(nik_ctx* c, Lane& L, const std::vector<nik_frame>& todo);
static int ensure_kzz(nik_ctx* c, int n, const nik_frame* keys) {
    std::vector<nik_frame> todo;
    for (int i = 0; i < n; ++i)
        if (!c->slot_kzz[keys[i]]) { c->slot_kzz[keys[i]] = 2; todo.push_back(keys[i]); }      // 2 = scheduled in this pass
    if (todo.empty()) return NIK_OK;
    Lane& L = c->lanes[0];
    int rc = ensure_kzz_run(c, L, todo);
    // whatever was scheduled but not enqueued (an error on the way) stays uncached
    for (nik_frame f : todo) if (c->slot_kzz[f] == 2) c->slot_kzz[f] = 0;
    return rc;
}
static int ensure_kzz_run(nik_ctx* c, Lane& L, const std::vector<nik_frame>& todo) {
    int rc;
    for (size_t b = 0; b < todo.size(); b += (size_t)c->max_batch) {
        const int m = (int)std::min(todo.size() - b, (size_t)c->max_batch);
        if ((rc = begin_call(c, L))) return rc;
        for (int i = 0; i < m; ++i) {
            if ((rc = depend_on_slot(c, L, 0, todo[b + i]))) return rc;
            note_read(c, L, 0, todo[b + i]);
            hidx(L, IX_KEY)[i] = todo[b + i];
        }
        if ((rc = upload_idx(c, L, IX_KEY, m))) return rc;
        const size_t item_stride = 2 * c->spec_max;
        for (int fam = 0; fam < 2; ++fam) {
            Family& f = fam ? c->img : c->pol;
            const float2* zsrc = fam ? c->arena_F : c->arena_P;
            float2* kz = fam ? c->arena_KzF : c->arena_KzP;
            unsigned* mz = fam ? c->arena_MzF : c->arena_MzP;
            if (c->cfg.kernel == 1)
                launch_energy(L.stream, m, f.g, zsrc, f.spec_elems, didx(L, IX_KEY), zsrc, f.spec_elems, didx(L, IX_KEY), L.energy);
            { Stage st(c, L, kname("kB", f.g.cols, "zz_inv", b_tag(c, f)).c_str(), m * 2 * Cb(f));
              launch_B_zz_inv(L.stream, m, f.g, f.t, zsrc, f.spec_elems, didx(L, IX_KEY), L.kbuf, item_stride, L.maxbuf); }
            { Stage st(c, L, kname("kA_inv", f.g.rows / 2, "kernel_fwd_z", a_tag(c, f)).c_str(), m * 2 * Cb(f));
              launch_A_inv_kernel_fwd(L.stream, m, f.g, f.t, L.kbuf, item_stride, c->spec_max, kernel_fn(c), L.maxbuf, L.energy, 0, 1); }
            { Stage st(c, L, kname("kB", f.g.cols, "fwd_kzz", b_tag(c, f)).c_str(), m * 2 * Cb(f));
              launch_B_fwd(L.stream, m, f.g, f.t, L.kbuf, item_stride, kz, f.spec_elems, didx(L, IX_KEY)); }
            launch_store_mzz(L.stream, m, f.g, L.maxbuf, didx(L, IX_KEY), mz);
        }
        HIP_TRY(c, hipGetLastError());
        // publish as a slot write of lane 0 so that other lanes order their reads after it
        L.write_seq += 1;
        HIP_TRY(c, hipEventRecord(L.write_ev, L.stream));
        for (int i = 0; i < m; ++i) { c->slot_lane[todo[b + i]] = 0; c->slot_seq[todo[b + i]] = L.write_seq; c->slot_kzz[todo[b + i]] = 1; }
        if ((rc = end_call(c, L))) return rc;
    }
    return NIK_OK;
}

// shared body of nik_pose_batch / nik_track_batch_dev
// upper != null: the arg-max windows are centred where `upper`'s latest pose call (same n, same stream split) found its peaks,
// predicted on the device (coarse-to-fine chaining).
static int pose_call(nik_ctx* c, int n, const uint8_t* d_gray, const nik_frame* keys, const nik_frame* curs,
                     int not_large_rotation, nik_pose_result* res, const int32_t* win_centers = nullptr, int win_radius = -1,
                     nik_ctx* upper = nullptr) {
    int rc;
    if (c->kzz_cache && (rc = ensure_kzz(c, n, keys))) return rc;
    const int nl = lanes_for(c, n);
    // lane rotation (nik_set_lane_rotation): the call's chunks go to the lanes behind those of the previous call, so that several
    // calls in flight -- a tracker's look-ahead batches -- run side by side instead of queueing on lane 0
    const bool rot = c->lane_rot && !upper && c->active_lanes > 1;
    const int LN = rot ? c->active_lanes : nl, lane_base = rot ? c->lane_base : 0;
    if ((rc = ensure_lanes(c, LN))) return rc;
    if (rot) c->lane_base = (lane_base + nl) % LN;
    // chunks: one per lane, or -- $NIK_CHUNK / nik_set_chunk -- pieces of at most chunk_pairs pairs dealt to the lanes in turn.
    // Small chunks keep a kernel's output in the 256 MiB Infinity Cache until the next kernel of the lane reads it.
    int nc = nl;
    if (c->chunk_pairs > 0 && !upper && n > nl * c->chunk_pairs) nc = ((n + c->chunk_pairs - 1) / c->chunk_pairs + nl - 1) / nl * nl;
    if (upper) {
        // the windows come from upper's latest pose call: it must have been a one-hypothesis call over the same n pairs,
        // cut into the same chunks (else the predicted centres would be stale or belong to other pairs)
        if (upper->last_pose_n != n || upper->last_pose_nhyp != 1 || upper->last_pose_nc != nc || lanes_for(upper, n) != nl)
            return fail(c, NIK_ERR_NOT_READY, "chained call: the upper level's latest pose call does not match (n %d/%d, hypotheses %d, chunks %d/%d)",
                        upper->last_pose_n, n, upper->last_pose_nhyp, upper->last_pose_nc, nc);
        // chunk ci of the upper call ran on lane (lane_base + ci) % LN when its lanes rotate (nik_set_lane_rotation): the windows
        // are read from upper->lanes[ci % nl], which is that lane only without rotation
        if (upper->lane_rot && upper->active_lanes > 1)
            return fail(c, NIK_ERR_NOT_READY, "chained call: the upper level rotates its lanes (nik_set_lane_rotation); its results cannot be chained");
    }
    if (c->want_stats && c->stats_parts + nc > KCC_STATS_PARTS) return fail(c, NIK_ERR_CAPACITY, "residual statistics: more than %d chunks in one call", (int)KCC_STATS_PARTS);
    c->last_pose_n = n; c->last_pose_nhyp = not_large_rotation ? 1 : 2; c->last_pose_nc = nc;
    for (int ci = 0; ci < nc; ++ci) {
        const int li = (lane_base + ci % nl) % LN;
        int b, e; chunk_of(n, nc, ci, b, e);
        const int m = e - b; if (m <= 0) continue;
        Lane& L = c->lanes[li];
        if ((rc = begin_call(c, L))) return rc;
        for (int i = b; i < e; ++i) {
            if ((rc = depend_on_slot(c, L, li, keys[i]))) return rc;
            note_read(c, L, li, keys[i]);
            if (d_gray) { if ((rc = depend_for_write(c, L, li, curs[i]))) return rc; }
            else { if ((rc = depend_on_slot(c, L, li, curs[i]))) return rc; note_read(c, L, li, curs[i]); }
        }
        // Small stored-frame batches are latency-bound by their ~16 dependent launches: replay them as one hipGraph (index
        // upload, kernels and result copies captured once per (ring entry, n, mode); all pointers are call-invariant).
        bool graphable = c->graph_max > 0 && !d_gray && !upper && !win_centers && nl == 1 && nc == 1 && m <= c->graph_max && !c->prof_on &&
                         !(c->want_stats && c->stats_parts != 0);
        if (graphable) for (int i = b; i < e; ++i) if (!(c->slot_kind[curs[i]] & 1)) graphable = false;
        Lane::PoseGraph* pg = nullptr;
        const int gslot = (int)(L.cur - L.ring), gflags = (not_large_rotation ? 1 : 0) | (c->want_stats ? 2 : 0) | (c->kzz_cache ? 4 : 0);
        if (graphable) {
            for (auto& g : L.graphs) if (g.slot == gslot && g.n == m && g.flags == gflags) pg = &g;
            // the first batch of a shape runs the ordinary way (it may initialise per-kernel launch attributes, which a
            // capture must not contain); the second one is captured
            if (!pg) { L.graphs.push_back({ gslot, m, gflags, nullptr, 0 }); graphable = false; }
        }
        if (graphable) {
            const int slot = gslot, flags = gflags;
            // fill the pinned index arrays (the captured copy node reads them at replay time)
            const int n_hyp = not_large_rotation ? 1 : 2;
            for (int i = 0; i < m; ++i) { hidx(L, IX_KEY)[i] = keys[b + i]; hidx(L, IX_CUR)[i] = curs[b + i]; }
            for (int t = 0; t < m * n_hyp; ++t) { hidx(L, IX_TIMG)[t] = curs[b + t / n_hyp]; hidx(L, IX_TKEY)[t] = keys[b + t / n_hyp]; }
            if (!pg->exec) {
                hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
                HIP_TRY(c, hipStreamBeginCapture(L.stream, hipStreamCaptureModeThreadLocal));
                L.idx_force = true;                          // the graph must contain the index upload whatever d_idx holds now
                rc = stage_pose_indices(c, L, m, keys + b, curs + b, not_large_rotation, false);
                L.idx_force = false;
                if (!rc) rc = enqueue_pose(c, L, m, not_large_rotation, true, false, -1);
                if (!rc && c->want_stats)
                    launch_residual_stats(L.stream, L.rot_res, L.trans_res, m, n_hyp, c->H, c->W, c->PD, c->PC, c->d_stats + 4 * (size_t)c->stats_parts);
                const hipError_t ce = hipStreamEndCapture(L.stream, &graph);
                if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
                HIP_TRY(c, ce);
                HIP_TRY(c, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                (void)hipGraphDestroy(graph);
                pg->exec = exec; (void)slot; (void)flags;
            }
            if (c->want_stats) {
                // (the captured statistics kernel writes the part slot it was captured with: graphs are only used with one chunk per call)
                if (c->stats_pending) HIP_TRY(c, hipStreamWaitEvent(L.stream, c->stats_done, 0));
                c->stats_parts += 1; c->stats_lanes = std::max(c->stats_lanes, li + 1);
            }
            HIP_TRY(c, hipGraphLaunch(pg->exec, L.stream));
            for (int w = 0; w < IX_ROTIDX; ++w) L.idx_shadow_n[w] = 0;      // (the replayed copy node rewrote d_idx behind the shadow's back)
            pg->uses += 1;
        } else {
            if ((rc = stage_pose_indices(c, L, m, keys + b, curs + b, not_large_rotation, d_gray != nullptr))) return rc;
            if (upper) {
                Lane& U = upper->lanes[li];
                HIP_TRY(c, hipStreamWaitEvent(L.stream, U.tail_ev, 0));       // the level above has enqueued its share already
                launch_predict_windows(L.stream, U.rot_res, U.trans_res, m, upper->PD, upper->PC, upper->H, upper->W, c->PD, c->PC, c->H, c->W,
                                       didx(L, IX_WRR), didx(L, IX_WRC), didx(L, IX_WTR), didx(L, IX_WTC));
                // (upper must not overwrite those results before this lane has read them)
                if (!upper->chain_ev[li]) HIP_TRY(c, hipEventCreateWithFlags(&upper->chain_ev[li], hipEventDisableTiming));
                HIP_TRY(c, hipEventRecord(upper->chain_ev[li], L.stream));
                upper->chain_pending[li] = true;
            }
            if (win_centers) {
                for (int i = 0; i < m; ++i) {
                    const int32_t* w = win_centers + 4 * (size_t)(b + i);
                    hidx(L, IX_WRR)[i] = w[0]; hidx(L, IX_WRC)[i] = w[1]; hidx(L, IX_WTR)[i] = w[2]; hidx(L, IX_WTC)[i] = w[3];
                }
                HIP_TRY(c, hipMemcpyAsync(L.d_idx + (size_t)L.cap_items * IX_WRR, L.cur->h_idx + (size_t)L.cap_items * IX_WRR,
                                          sizeof(int) * (size_t)L.cap_items * 4, hipMemcpyHostToDevice, L.stream));
            }
            bool fuse = false, img_u8 = true;
            if (d_gray) {
                // the polar spectrum's last pass is fused into the pose's first kernel (not for the gaussian kernel,
                // which needs sum|X|^2 of the finished spectrum before that kernel runs)
                fuse = (c->cfg.kernel != 1) && c->fuse_polar;
                enqueue_intermedium(c, L, m, d_gray + (size_t)b * c->img.real_elems, fuse);
                // (fused: the frames' polar spectra are completed by the pose's first kernel -- the write event other
                // lanes wait on must come after it)
                if (!fuse && (rc = mark_written(c, L, li, curs + b, m, 1))) return rc;
            } else {
                // stored frames: de-rotate from the u8 frame store when every current frame has a u8 image; a batch that
                // mixes in f32 frames (nik_intermedium_f32 / nik_frame_import) runs on f32 planes, materialised on demand
                for (int i = b; i < e; ++i) if (!(c->slot_kind[curs[i]] & 1)) img_u8 = false;
                if (!img_u8 && (rc = ensure_f32_images(c, L, li, m, curs + b))) return rc;
            }
            if ((rc = enqueue_pose(c, L, m, not_large_rotation, img_u8, fuse, (win_centers || upper) ? win_radius : -1))) return rc;
            if (c->want_stats) {                                  // this lane's share of the batch's residual statistics
                if (c->stats_pending) HIP_TRY(c, hipStreamWaitEvent(L.stream, c->stats_done, 0));   // (the previous batch's partials are consumed)
                launch_residual_stats(L.stream, L.rot_res, L.trans_res, m, not_large_rotation ? 1 : 2, c->H, c->W, c->PD, c->PC, c->d_stats + 4 * (size_t)c->stats_parts);
                c->stats_parts += 1; c->stats_lanes = std::max(c->stats_lanes, li + 1);
            }
            if (fuse && (rc = mark_written(c, L, li, curs + b, m, 1))) return rc;
        }
        HIP_TRY(c, hipGetLastError());
        L.cur->has_pose = true; L.cur->n = m; L.cur->n_hyp = not_large_rotation ? 1 : 2; L.cur->res = res ? res + b : nullptr;
        if ((rc = end_call(c, L))) return rc;
    }
    return NIK_OK;
}

int nik_pose_batch_async(nik_ctx* c, int n, const nik_frame* keys, const nik_frame* curs, int not_large_rotation,
                         nik_pose_result* res) {
    if (!c || !keys || !curs || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    int rc;
    if ((rc = check_kernel(c))) return rc;
    if (n == 0) return NIK_OK;
    if (n > c->max_batch) return fail(c, NIK_ERR_CAPACITY, "batch %d exceeds max_batch %d", n, c->max_batch);
    for (int i = 0; i < n; ++i) if ((rc = check_slot(c, keys[i], true)) || (rc = check_slot(c, curs[i], true))) return rc;
    c->stats_parts = 0; c->stats_lanes = 0;
    return pose_call(c, n, nullptr, keys, curs, not_large_rotation, res);
}

int nik_pose_batch(nik_ctx* c, int n, const nik_frame* keys, const nik_frame* curs, int not_large_rotation,
                   nik_pose_result* res) {
    int rc = nik_pose_batch_async(c, n, keys, curs, not_large_rotation, res);
    if (rc || n <= 0) return rc;
    return drain_all(c);
}

// Results of one asynchronous batch: waits for (and finalises) every in-flight call that writes into res[0, n), and -- calls of a
// lane retire in order -- whatever that lane was given before them.  Later calls keep running.
int nik_wait_results(nik_ctx* c, const nik_pose_result* res, int n) {
    if (!c || !res || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    for (int li = 0; li < c->active_lanes; ++li) {
        Lane& L = c->lanes[li];
        int last = -1;
        for (int k = 0; k < L.ring_n; ++k) {        // oldest call first
            const Call& call = L.ring[(L.next + k) % L.ring_n];
            if (call.busy && call.res && call.res >= res && call.res < res + n) last = k;
        }
        for (int k = 0; k <= last; ++k) {
            int rc = retire(c, L.ring[(L.next + k) % L.ring_n]);
            if (rc) return rc;
        }
    }
    return NIK_OK;
}

int nik_set_lane_rotation(nik_ctx* c, int on) {
    if (!c) return NIK_ERR_INVALID_ARG;
    c->lane_rot = on != 0;
    return NIK_OK;
}

int nik_pose_batch_window(nik_ctx* c, int n, const nik_frame* keys, const nik_frame* curs, const int32_t* centers, int radius,
                          nik_pose_result* res) {
    if (!c || !keys || !curs || !centers || n < 0 || radius < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    int rc;
    if ((rc = check_kernel(c))) return rc;
    if (n == 0) return NIK_OK;
    if (n > c->max_batch) return fail(c, NIK_ERR_CAPACITY, "batch %d exceeds max_batch %d", n, c->max_batch);
    for (int i = 0; i < n; ++i) {
        if ((rc = check_slot(c, keys[i], true)) || (rc = check_slot(c, curs[i], true))) return rc;
        const int32_t* w = centers + 4 * (size_t)i;
        if (w[0] < 0 || w[0] >= c->PD || w[1] < 0 || w[1] >= c->PC || w[2] < 0 || w[2] >= c->H || w[3] < 0 || w[3] >= c->W)
            return fail(c, NIK_ERR_INVALID_ARG, "window centre of pair %d outside its surface", i);
    }
    c->stats_parts = 0; c->stats_lanes = 0;
    if ((rc = pose_call(c, n, nullptr, keys, curs, 1, res, centers, radius))) return rc;
    return drain_all(c);
}

// Coarse-to-fine chaining: nik_pose_batch_window whose window centres come, on the device, from `upper`'s latest pose call
// over the same n pairs.  Asynchronous unless sync.
int nik_pose_batch_chained(nik_ctx* c, int n, const nik_frame* keys, const nik_frame* curs, nik_ctx* upper, int radius,
                           nik_pose_result* res, int sync) {
    if (!c || !upper || !keys || !curs || n < 0 || radius < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    if (upper->device != c->device) return fail(c, NIK_ERR_INVALID_ARG, "chained levels must live on one device");
    int rc;
    if ((rc = check_kernel(c))) return rc;
    if (n == 0) return NIK_OK;
    if (n > c->max_batch) return fail(c, NIK_ERR_CAPACITY, "batch %d exceeds max_batch %d", n, c->max_batch);
    for (int i = 0; i < n; ++i) if ((rc = check_slot(c, keys[i], true)) || (rc = check_slot(c, curs[i], true))) return rc;
    c->stats_parts = 0; c->stats_lanes = 0;
    if ((rc = pose_call(c, n, nullptr, keys, curs, 1, res, nullptr, radius, upper))) return rc;
    return sync ? drain_all(c) : NIK_OK;
}

// every stream of c waits for the work `other` (same device) has enqueued so far
int nik_wait_for(nik_ctx* c, nik_ctx* other) {
    if (!c || !other) return NIK_ERR_INVALID_ARG;
    if (other->device != c->device) return fail(c, NIK_ERR_INVALID_ARG, "contexts on different devices");
    if (!other->fence_ev) HIP_TRY(c, hipEventCreateWithFlags(&other->fence_ev, hipEventDisableTiming));
    int rc = ensure_lanes(c, c->active_lanes);
    if (rc) return rc;
    for (int lo = 0; lo < other->active_lanes; ++lo) {
        if (!other->lanes[lo].stream) continue;
        HIP_TRY(c, hipEventRecord(other->fence_ev, other->lanes[lo].stream));
        for (int li = 0; li < c->active_lanes; ++li) HIP_TRY(c, hipStreamWaitEvent(c->lanes[li].stream, other->fence_ev, 0));
    }
    return NIK_OK;
}

// Stream plumbing for callers that chain several contexts without returning to the host (kcc_pyramid.cpp).
int nik_set_call_depth(nik_ctx* c, int depth) {
    if (!c || depth < 1 || depth > KCC_RING_MAX) return fail(c, NIK_ERR_INVALID_ARG, "call depth must be 1..%d", (int)KCC_RING_MAX);
    int rc = drain_all(c);
    if (rc) return rc;
    for (Lane& L : c->lanes) { L.ring_n = depth; L.next = 0; }
    return NIK_OK;
}
// `stream` (a hipStream_t of the same device) waits for everything c has enqueued so far
int nik_stream_wait_ctx(nik_ctx* c, void* stream) {
    if (!c) return NIK_ERR_INVALID_ARG;
    if (!c->fence_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->fence_ev, hipEventDisableTiming));
    for (int li = 0; li < c->active_lanes; ++li) {
        if (!c->lanes[li].stream) continue;                  // (never used: nothing to wait for)
        HIP_TRY(c, hipEventRecord(c->fence_ev, c->lanes[li].stream));
        HIP_TRY(c, hipStreamWaitEvent((hipStream_t)stream, c->fence_ev, 0));
    }
    return NIK_OK;
}
// every stream of c waits for what has been enqueued on `stream` so far
int nik_ctx_wait_stream(nik_ctx* c, void* stream) {
    if (!c) return NIK_ERR_INVALID_ARG;
    if (!c->fence_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->fence_ev, hipEventDisableTiming));
    int rc = ensure_lanes(c, c->active_lanes);
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->fence_ev, (hipStream_t)stream));
    for (int li = 0; li < c->active_lanes; ++li) HIP_TRY(c, hipStreamWaitEvent(c->lanes[li].stream, c->fence_ev, 0));
    return NIK_OK;
}
// 2x2 box filter of n frames of c's geometry on a caller-owned stream
int nik_downsample_u8_stream(nik_ctx* c, int n, const uint8_t* d_in, uint8_t* d_out, void* stream) {
    if (!c || !d_in || !d_out || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    if (n == 0) return NIK_OK;
    launch_downsample_u8((hipStream_t)stream, n, d_in, d_out, c->H, c->W);
    HIP_TRY(c, hipGetLastError());
    return NIK_OK;
}

// `steps` (1..3) pyramid levels below ctx's geometry in ONE launch on `stream`: frames [0, na) of d_a, then nb frames of d_b
// (may be null with nb = 0); out[d] receives the na + nb frames of level d + 1, back to back.  The same integers as `steps`
// chained nik_downsample_u8 calls.  Needs H, W divisible by 2^steps and pointers aligned to 2^steps bytes, else
// NIK_ERR_UNSUPPORTED_SIZE (the caller falls back to the chained form).
int nik_downsample_pyr_u8_stream(nik_ctx* c, int steps, int na, const uint8_t* d_a, int nb, const uint8_t* d_b, uint8_t* const* out, void* stream) {
    if (!c || !d_a || !out || steps < 1 || steps > 3 || na < 0 || nb < 0 || (nb > 0 && !d_b)) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    const uintptr_t m = (uintptr_t)(1 << steps) - 1;
    uintptr_t bits = (uintptr_t)d_a | (uintptr_t)(nb ? d_b : nullptr) | (uintptr_t)c->H | (uintptr_t)c->W;
    for (int d = 0; d < steps; ++d) { if (!out[d]) return fail(c, NIK_ERR_INVALID_ARG, "null output"); bits |= (uintptr_t)out[d] << (d + 1); }
    if (bits & m) return fail(c, NIK_ERR_UNSUPPORTED_SIZE, "fused downsample needs sizes and pointers aligned to %d", 1 << steps);
    if (na + nb == 0) return NIK_OK;
    launch_downsample_pyr((hipStream_t)stream, steps, d_a, d_b, na, na + nb, c->H, c->W, out);
    HIP_TRY(c, hipGetLastError());
    return NIK_OK;
}

// nik_downsample_u8_dev without the drain: enqueued on nik_stream(ctx)
int nik_downsample_u8_async(nik_ctx* c, int n, const uint8_t* d_in, uint8_t* d_out) {
    if (!c || !d_in || !d_out || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    if (n == 0) return NIK_OK;
    launch_downsample_u8(c->lanes[0].stream, n, d_in, d_out, c->H, c->W);
    HIP_TRY(c, hipGetLastError());
    return NIK_OK;
}

int nik_downsample_u8_dev(nik_ctx* c, int n, const uint8_t* d_in, uint8_t* d_out) {
    if (!c || !d_in || !d_out || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    if (n == 0) return NIK_OK;
    int rc = drain_all(c);
    if (rc) return rc;
    launch_downsample_u8(c->lanes[0].stream, n, d_in, d_out, c->H, c->W);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->lanes[0].stream));
    return NIK_OK;
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
    for (int i = 0; i < n; ++i) if ((rc = check_slot(c, keys[i], true)) || (rc = check_slot(c, cur_dst[i], false))) return rc;
    // a slot that is both read as a key and overwritten, or overwritten twice, would make the results depend on how the
    // batch is split over the streams: reject it
    {
        std::vector<uint8_t> mark(c->max_frames, 0);
        for (int i = 0; i < n; ++i) mark[keys[i]] = 1;
        for (int i = 0; i < n; ++i) {
            if (mark[cur_dst[i]] == 1) return fail(c, NIK_ERR_INVALID_ARG, "slot %d is both a key and a destination of the batch", cur_dst[i]);
            if (mark[cur_dst[i]] == 2) return fail(c, NIK_ERR_INVALID_ARG, "slot %d is a destination twice in the batch", cur_dst[i]);
            mark[cur_dst[i]] = 2;
        }
    }
    c->stats_parts = 0; c->stats_lanes = 0;
    if ((rc = pose_call(c, n, d_gray, keys, cur_dst, not_large_rotation, res))) return rc;
    return sync ? drain_all(c) : NIK_OK;
}

int nik_rgb_to_gray_dev(nik_ctx* c, int n, const uint8_t* d_rgb, int bgr, uint8_t* d_gray) {
    if (!c || !d_rgb || !d_gray || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    if (n == 0) return NIK_OK;
    // runs on lane 0 and is ordered before later calls by a full drain (colour conversion is not on the timed path)
    int rc = drain_all(c);
    if (rc) return rc;
    launch_rgb2gray(c->lanes[0].stream, d_rgb, d_gray, (size_t)n * c->img.real_elems, bgr);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->lanes[0].stream));
    return NIK_OK;
}

// nik_rgb_to_gray_dev without the host synchronisation: converted on the first stream, every stream of the context waits
// for it on the device; d_gray must not still be read by calls enqueued earlier (use two buffers, or the same buffer
// only after nik_synchronize) -- stream order covers calls enqueued LATER
int nik_rgb_to_gray_async(nik_ctx* c, int n, const uint8_t* d_rgb, int bgr, uint8_t* d_gray) {
    if (!c || !d_rgb || !d_gray || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    if (n == 0) return NIK_OK;
    if (!c->fence_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->fence_ev, hipEventDisableTiming));
    { int rce = ensure_lanes(c, c->active_lanes); if (rce) return rce; }
    // the conversion overwrites d_gray: it must come after whatever the other streams still read from it
    for (int li = 1; li < c->active_lanes; ++li) {
        HIP_TRY(c, hipEventRecord(c->fence_ev, c->lanes[li].stream));
        HIP_TRY(c, hipStreamWaitEvent(c->lanes[0].stream, c->fence_ev, 0));
    }
    launch_rgb2gray(c->lanes[0].stream, d_rgb, d_gray, (size_t)n * c->img.real_elems, bgr);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->fence_ev, c->lanes[0].stream));
    for (int li = 1; li < c->active_lanes; ++li) HIP_TRY(c, hipStreamWaitEvent(c->lanes[li].stream, c->fence_ev, 0));
    return NIK_OK;
}

int nik_match(nik_ctx* c, nik_frame query, int n, const nik_frame* cands, int* best, nik_pose_result* res,
              nik_pose_result* best_res) {
    if (!c || (n > 0 && !cands) || n < 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    if (best) *best = -1;
    if (n == 0) return NIK_OK;
    std::vector<nik_pose_result> local;
    if (!res) { local.resize(n); res = local.data(); }
    int rc;
    if ((rc = check_kernel(c))) return rc;
    for (int i = 0; i < n; ++i) if ((rc = check_slot(c, cands[i], true))) return rc;
    if ((rc = check_slot(c, query, true))) return rc;
    // ComputePose(cand_i, query, not_large_rotation=false) for every candidate (loop_closure.cc:58-59), max_batch at a time;
    // the chunks are queued back to back (two in flight per lane) and finalised by the drain
    std::vector<nik_frame> curs(std::min(n, c->max_batch), query);
    c->stats_parts = 0; c->stats_lanes = 0;      // the statistics cover every chunk of this search
    for (int b = 0; b < n; b += c->max_batch) {
        const int m = std::min(c->max_batch, n - b);
        if ((rc = pose_call(c, m, nullptr, cands + b, curs.data(), 0, res + b))) return rc;
    }
    if ((rc = drain_all(c))) return rc;
    int bi = -1; double bs = -3.0;                                   // LoopClosureResult(): response(-1,-1,-1)  (loop_closure.h:14)
    for (int i = 0; i < n; ++i) {
        const double s = res[i].info[0] + res[i].info[1] + res[i].info[2];
        if (s > bs) { bs = s; bi = i; }                              // loop_closure.cc:61 strict >
    }
    if (best) *best = bi;
    if (best_res && bi >= 0) *best_res = res[bi];
    return NIK_OK;
}

// rotation stage only (EstimateTrans on the cached polar spectra): PSR_r and arg-max row per candidate
static int rotation_call(nik_ctx* c, int n, const nik_frame* keys, const nik_frame* curs, float* psr_rot, int* rot_row) {
    int rc;
    if (c->kzz_cache && (rc = ensure_kzz(c, n, keys))) return rc;      // enqueue_estimate's cached branch reads the keys' Kzz
    struct Part { Lane* L; Call* call; int b, m; };
    std::vector<Part> parts;
    const int nl = lanes_for(c, n);
    if ((rc = ensure_lanes(c, nl))) return rc;
    for (int li = 0; li < nl; ++li) {
        int b, e; chunk_of(n, nl, li, b, e);
        const int m = e - b; if (m <= 0) continue;
        Lane& L = c->lanes[li];
        if ((rc = begin_call(c, L))) return rc;
        for (int i = b; i < e; ++i) {
            if ((rc = depend_on_slot(c, L, li, keys[i])) || (rc = depend_on_slot(c, L, li, curs[i]))) return rc;
            note_read(c, L, li, keys[i]); note_read(c, L, li, curs[i]);
        }
        if ((rc = stage_pose_indices(c, L, m, keys + b, curs + b, 1, false))) return rc;
        L.mirror = L.cur->h_rot;
        enqueue_estimate(c, L, m, c->pol, false, c->arena_P, c->pol.spec_elems, didx(L, IX_CUR),
                         c->arena_P, c->pol.spec_elems, didx(L, IX_KEY), L.rot_res, nullptr, 1);
        HIP_TRY(c, hipGetLastError());
        L.cur->has_pose = false;
        if ((rc = end_call(c, L))) return rc;
        parts.push_back({ &L, L.cur, b, m });
    }
    for (const Part& p : parts) {
        HIP_TRY(c, hipEventSynchronize(p.call->done));
        for (int i = 0; i < p.m; ++i) {
            psr_rot[p.b + i] = psr_from(p.call->h_rot[i], (long)c->PD * c->PC);
            rot_row[p.b + i] = p.call->h_rot[i].idx % c->PD;
        }
    }
    return NIK_OK;
}

int nik_match_topk(nik_ctx* c, nik_frame query, int n, const nik_frame* cands, int k, int* best,
                   nik_pose_result* best_res, int* shortlist) {
    if (!c || (n > 0 && !cands) || n < 0 || k <= 0) return fail(c, NIK_ERR_INVALID_ARG, "null/negative argument");
    if (best) *best = -1;
    if (n == 0) return NIK_OK;
    int rc;
    if ((rc = check_kernel(c)) || (rc = check_slot(c, query, true))) return rc;
    for (int i = 0; i < n; ++i) if ((rc = check_slot(c, cands[i], true))) return rc;
    if ((rc = drain_all(c))) return rc;
    // stage 1: rank every candidate by the rotation-stage PSR (needs only the cached polar spectra)
    std::vector<float> psr(n); std::vector<int> row(n);
    std::vector<nik_frame> curs(std::min(n, c->max_batch), query);
    for (int b = 0; b < n; b += c->max_batch) {
        const int m = std::min(c->max_batch, n - b);
        if ((rc = rotation_call(c, m, cands + b, curs.data(), psr.data() + b, row.data() + b))) return rc;
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    k = std::min(k, n);
    std::partial_sort(order.begin(), order.begin() + k, order.end(),
                      [&](int a, int b2) { return psr[a] > psr[b2] || (psr[a] == psr[b2] && a < b2); });
    std::sort(order.begin(), order.begin() + k);              // keep candidate order: ties resolve as in the exact search
    // stage 2: the reference's full two-hypothesis ComputePose on the short list only
    std::vector<nik_frame> top(k);
    for (int i = 0; i < k; ++i) { top[i] = cands[order[i]]; if (shortlist) shortlist[i] = order[i]; }
    std::vector<nik_pose_result> res(k);
    int bi = -1;
    if ((rc = nik_match(c, query, k, top.data(), &bi, res.data(), best_res))) return rc;
    if (best) *best = bi >= 0 ? order[bi] : -1;
    return NIK_OK;
}

int nik_profile_enable(nik_ctx* c, int enable) {
    if (!c) return NIK_ERR_INVALID_ARG;
    int rc = nik_synchronize(c);
    if (rc) return rc;
    for (auto& r : c->prof_recs) { c->prof_pool.push_back(r.a); c->prof_pool.push_back(r.b); }
    c->prof_recs.clear(); c->prof_stats.clear();
    c->prof_on = enable != 0;
    return NIK_OK;
}

int nik_profile_read(nik_ctx* c, nik_stage_stat* out, int cap, int* n) {
    if (!c || !n) return NIK_ERR_INVALID_ARG;
    int rc = nik_synchronize(c);
    if (rc) return rc;
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
        out[i].ms = c->prof_stats[i].ms; out[i].launches = c->prof_stats[i].launches; out[i].bytes = c->prof_stats[i].bytes; out[i].bytes_design = c->prof_stats[i].bytes_design;
    }
    return NIK_OK;
}

// ---- debug taps ---------------------------------------------------------------------------------

int nik_dbg_fft(nik_ctx* c, int which, const float* x, float* xf_out) {
    if (!c || !x || !xf_out) return NIK_ERR_INVALID_ARG;
    int rc;
    if ((rc = nik_synchronize(c))) return rc;
    Family& f = which ? c->pol : c->img;
    Lane& L = c->lanes[0]; hipStream_t s = L.stream;
    float* d_in = c->d_scratch;                                                   // first half: real input
    float2* d_out = reinterpret_cast<float2*>(c->d_scratch) + c->spec_max;        // second half: transposed output
    HIP_TRY(c, hipMemcpyAsync(d_in, x, sizeof(float) * f.real_elems, hipMemcpyHostToDevice, s));
    if (which ? c->gen_pol : c->gen_img) {
        g_rfft2(s, 1, which ? c->gpol : c->gimg, d_in, f.real_elems, f.g.rows, nullptr, L.tmpA, c->spec_max, nullptr);
    } else {
    launch_A_fwd_plane(s, 1, f.g, f.t, d_in, f.real_elems, f.g.rows, nullptr, L.tmpA, c->spec_max);
    launch_B_fwd(s, 1, f.g, f.t, L.tmpA, c->spec_max, L.tmpA, c->spec_max, nullptr);
    }
    launch_transpose_c(s, L.tmpA, d_out, f.g.hr, f.g.cols);
    HIP_TRY(c, hipMemcpyAsync(xf_out, d_out, sizeof(float2) * f.spec_elems, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    return NIK_OK;
}

int nik_dbg_ifft(nik_ctx* c, int which, const float* xf, float* x_out) {
    if (!c || !xf || !x_out) return NIK_ERR_INVALID_ARG;
    int rc;
    if ((rc = nik_synchronize(c))) return rc;
    Family& f = which ? c->pol : c->img;
    Lane& L = c->lanes[0]; hipStream_t s = L.stream;
    float2* scratch = reinterpret_cast<float2*>(c->d_scratch);
    HIP_TRY(c, hipMemcpyAsync(scratch, xf, sizeof(float2) * f.spec_elems, hipMemcpyHostToDevice, s));
    launch_transpose_c(s, scratch, L.tmpA, f.g.cols, f.g.hr);
    float* dst = reinterpret_cast<float*>(L.kbuf);
    if (which ? c->gen_pol : c->gen_img) {
        g_irfft2(s, 1, which ? c->gpol : c->gimg, L.tmpA, c->spec_max, nullptr, dst, f.real_elems, f.g.rows);
    } else {
    launch_B_inv(s, 1, f.g, f.t, L.tmpA, c->spec_max, L.gbuf, c->spec_max);
    launch_A_inv_real(s, 1, f.g, f.t, L.gbuf, c->spec_max, dst, f.real_elems);
    }
    HIP_TRY(c, hipMemcpyAsync(x_out, dst, sizeof(float) * f.real_elems, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    return NIK_OK;
}

// RotateArray(frame image, degree2 / 2 degrees) through the hot gather (u8 or f32 source, whichever the slot holds)
int nik_dbg_rotate(nik_ctx* c, nik_frame fr, int degree2, float* out) {
    if (!c || !out) return NIK_ERR_INVALID_ARG;
    int rc;
    if ((rc = nik_synchronize(c)) || (rc = check_slot(c, fr, false))) return rc;
    if (!(c->slot_ready[fr] & 1)) return fail(c, NIK_ERR_NOT_READY, "frame slot %d holds no image", fr);
    Lane& L = c->lanes[0]; hipStream_t s = L.stream;
    std::vector<int> terms((size_t)2 * c->W + 2 * c->H + 2);
    rotation_terms(c->H, c->W, (float)degree2 * 0.5f, terms.data());             // RotateArray(image, degree2/2)
    terms[terms.size() - 2] = fr; terms[terms.size() - 1] = 0;                   // [slot, table index] behind the terms
    if (!c->rot_one) HIP_TRY(c, hipMalloc(&c->rot_one, sizeof(int) * terms.size()));
    HIP_TRY(c, hipMemcpyAsync(c->rot_one, terms.data(), sizeof(int) * terms.size(), hipMemcpyHostToDevice, s));
    const int* d_slot = c->rot_one + terms.size() - 2; const int* d_index = d_slot + 1;
    if (c->gen_img)
        g_rotate(s, 1, (c->slot_kind[fr] & 1) ? c->arena_u8 : nullptr, c->u8_stride, c->u8_pitch, c->arena_img, c->img_stride, c->img_pitch, d_slot, c->rot_one, d_index,
                 c->d_scratch, c->img.real_elems, c->H, c->W);
    else if (c->slot_kind[fr] & 1)
        launch_A_fwd_rot8(s, 1, c->img.g, c->img.t, c->arena_u8, c->u8_stride, c->u8_pitch, d_slot, c->rot_one, d_index, L.tmpA, c->spec_max, c->d_scratch);
    else
        launch_A_fwd_rot(s, 1, c->img.g, c->img.t, c->arena_img, c->img_stride, c->img_pitch, d_slot, c->rot_one, d_index, L.tmpA, c->spec_max, c->d_scratch);
    HIP_TRY(c, hipMemcpyAsync(out, c->d_scratch, sizeof(float) * c->img.real_elems, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    return NIK_OK;
}

// polar(fftshift(RemoveZeroComponent(x))) through the hot gather
int nik_dbg_polar(nik_ctx* c, const float* x, float* out) {
    if (!c || !x || !out) return NIK_ERR_INVALID_ARG;
    int rc;
    if ((rc = nik_synchronize(c))) return rc;
    Lane& L = c->lanes[0]; hipStream_t s = L.stream;
    float* d_out = reinterpret_cast<float*>(L.gbuf);
    float* d_in = c->d_scratch;
    HIP_TRY(c, hipMemcpyAsync(d_in, x, sizeof(float) * c->img.real_elems, hipMemcpyHostToDevice, s));
    // (the shifted plane by the plain kernels -- any size --, the gather by the polar family's own kernel)
    launch_make_shifted(s, d_in, L.splane, c->H, c->W);
    launch_fix_zero(s, 1, L.splane, c->s_elems, c->H, c->W);
    if (c->gen_pol) g_polar(s, 1, L.splane, c->s_elems, c->g_polar_map, d_out, c->pol.real_elems, c->H, c->PD, c->PC);
    else launch_A_fwd_polar(s, 1, c->pol.g, c->pol.t, L.splane, c->s_elems, c->H, c->W, c->polar, L.tmpA, c->spec_max, d_out);
    HIP_TRY(c, hipMemcpyAsync(out, d_out, sizeof(float) * c->pol.real_elems, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    return NIK_OK;
}

// The response surface g = IFFT(G) of one EstimateTrans call (correlation_flow.cc:171-173) -- the plane whose arg-max and
// moments the hot path reduces on registers without ever storing it.  Tests use it to MEASURE the float32 noise between
// this implementation and the oracle (the rotation-tie tolerance is derived from that measurement).
//   which 0: rotation stage, z = key's polar spectrum, x = cur's polar spectrum          -> PD x PC (column-major)
//   which 1: translation stage, z = key's spectrum, x = FFT(RotateArray(cur image, degree2 / 2 degrees)) -> H x W
int nik_dbg_response(nik_ctx* c, int which, nik_frame key, nik_frame cur, int degree2, float* g) {
    if (!c || !g || (which != 0 && which != 1)) return NIK_ERR_INVALID_ARG;
    int rc;
    if ((rc = check_kernel(c)) || (rc = nik_synchronize(c)) || (rc = check_slot(c, key, true)) || (rc = check_slot(c, cur, true))) return rc;
    if (c->kzz_cache && (rc = ensure_kzz(c, 1, &key))) return rc;
    Lane& L = c->lanes[0]; hipStream_t s = L.stream;
    if ((rc = begin_call(c, L)) || (rc = depend_on_slot(c, L, 0, key)) || (rc = depend_on_slot(c, L, 0, cur))) return rc;
    if ((rc = stage_pose_indices(c, L, 1, &key, &cur, 1, false))) return rc;
    Family& f = which ? c->img : c->pol;
    if (which == 0) {
        enqueue_estimate(c, L, 1, c->pol, false, c->arena_P, c->pol.spec_elems, didx(L, IX_CUR),
                         c->arena_P, c->pol.spec_elems, didx(L, IX_KEY), L.rot_res, nullptr, 1);
    } else {
        std::vector<int> terms((size_t)2 * c->W + 2 * c->H);
        rotation_terms(c->H, c->W, (float)degree2 * 0.5f, terms.data());
        if (!c->rot_one) HIP_TRY(c, hipMalloc(&c->rot_one, sizeof(int) * (terms.size() + 2)));
        HIP_TRY(c, hipMemcpyAsync(c->rot_one, terms.data(), sizeof(int) * terms.size(), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemsetAsync(didx(L, IX_ROTIDX), 0, sizeof(int), s));
        if (!(c->slot_kind[cur] & 1) && (rc = ensure_f32_images(c, L, 0, 1, &cur))) return rc;
        if (c->gen_img) {
            g_rotate(s, 1, (c->slot_kind[cur] & 1) ? c->arena_u8 : nullptr, c->u8_stride, c->u8_pitch, c->arena_img, c->img_stride, c->img_pitch, didx(L, IX_TIMG),
                     c->rot_one, didx(L, IX_ROTIDX), L.rbuf, 2 * c->r_elems, c->H, c->W);
            g_rfft2(s, 1, c->gimg, L.rbuf, 2 * c->r_elems, c->H, nullptr, L.tmpA, c->spec_max, nullptr);
            enqueue_estimate(c, L, 1, c->img, false, L.tmpA, c->spec_max, nullptr, c->arena_F, c->img.spec_elems, didx(L, IX_TKEY), L.trans_res, nullptr, 1);
        } else if (c->slot_kind[cur] & 1)
            launch_A_fwd_rot8(s, 1, c->img.g, c->img.t, c->arena_u8, c->u8_stride, c->u8_pitch, didx(L, IX_TIMG), c->rot_one, didx(L, IX_ROTIDX), L.tmpA, c->spec_max);
        else
            launch_A_fwd_rot(s, 1, c->img.g, c->img.t, c->arena_img, c->img_stride, c->img_pitch, didx(L, IX_TIMG), c->rot_one, didx(L, IX_ROTIDX), L.tmpA, c->spec_max);
        if (c->gen_img) {
            // (done above)
        } else if (c->cfg.kernel == 1) {
            launch_B_fwd(s, 1, c->img.g, c->img.t, L.tmpA, c->spec_max, L.tmpA, c->spec_max, nullptr);
            enqueue_estimate(c, L, 1, c->img, false, L.tmpA, c->spec_max, nullptr, c->arena_F, c->img.spec_elems, didx(L, IX_TKEY), L.trans_res, nullptr, 1);
        } else {
            enqueue_estimate(c, L, 1, c->img, true, L.tmpA, c->spec_max, nullptr, c->arena_F, c->img.spec_elems, didx(L, IX_TKEY), L.trans_res, nullptr, 1);
        }
    }
    float* d_g = reinterpret_cast<float*>(L.kbuf);            // (the kernel planes are consumed by now)
    if (which ? c->gen_img : c->gen_pol) d_g = L.rbuf;        // the any-size kernels materialise g (item 0, plane 0) before their arg-max
    else launch_A_inv_real(s, 1, f.g, f.t, L.gbuf, c->spec_max, d_g, f.real_elems);
    HIP_TRY(c, hipMemcpyAsync(g, d_g, sizeof(float) * f.real_elems, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipGetLastError());
    L.cur->has_pose = false;
    if ((rc = end_call(c, L))) return rc;
    return nik_synchronize(c);
}

// ---- host-side gather tables (tests; no device needed) -------------------------------------------

int nik_host_polar_plan(int H, int W, int PD, int PC, int dims[8], uint32_t** chunks, int* n_chunks, int** seg_first, uint32_t** pts) {
    if (!dims || !chunks || !n_chunks || !seg_first || !pts) return NIK_ERR_INVALID_ARG;
    if (!fft_half_supported(PD / 2)) return fail(nullptr, NIK_ERR_UNSUPPORTED_SIZE, "FFT length not instantiated");
    const FwdGeom fg = fwd_geom(PD / 2);
    PolarPlanHost h; std::string err;
    const char* al = kcc::tune_env("NIK_POLAR_ALIGNED");
    if (build_polar_plan(H, W, PD, PC, fg.lines, fg.threads, fg.rf, fg.mf, fg.lds_bytes, fg.qs_opts, h, err, al ? atoi(al) : KCC_POLAR_ALIGNED_DEFAULT)) return fail(nullptr, NIK_ERR_UNSUPPORTED_SIZE, "%s", err.c_str());
    const int d[8] = { h.qs, h.nseg, h.tiles, h.lines, h.threads, h.rf, h.mf, (int)h.lds_bytes };
    memcpy(dims, d, sizeof(d));
    auto dup = [](const void* src, size_t bytes) { void* p = malloc(std::max<size_t>(bytes, 1)); if (p) memcpy(p, src, bytes); return p; };
    *chunks = (uint32_t*)dup(h.chunks.data(), h.chunks.size() * 4); *n_chunks = (int)h.chunks.size();
    *seg_first = (int*)dup(h.seg_first.data(), h.seg_first.size() * 4);
    *pts = (uint32_t*)dup(h.pts.data(), h.pts.size() * 4);
    return (*chunks && *seg_first && *pts) ? NIK_OK : NIK_ERR_INVALID_ARG;
}
void nik_host_free(void* p) { free(p); }
int nik_host_rot_terms(int H, int W, float degree, int* out) {
    if (!out || H <= 0 || W <= 0) return NIK_ERR_INVALID_ARG;
    rotation_terms(H, W, degree, out);
    return NIK_OK;
}
// the run-time FFT plan the any-size kernels use for a line of n points: its radices in execution order (CPU tests)
int nik_host_fft_plan(int n, int radices[16]) {
    if (n < 1 || !radices) return NIK_ERR_INVALID_ARG;
    const GPlan p = gplan_make(n, nullptr);
    for (int i = 0; i < 16; ++i) radices[i] = i < p.nr ? p.radix[i] : 0;
    return p.nr;
}
int nik_host_rot8_geom(int H, int geom[5]) {
    if (!geom || !fft_half_supported(H / 2)) return NIK_ERR_INVALID_ARG;
    const Rot8Geom r = rot8_geom(H / 2);
    geom[0] = r.band_rows; geom[1] = r.bands; geom[2] = r.box_rows; geom[3] = r.pitch; geom[4] = r.lds_bytes;
    return NIK_OK;
}

}  // extern "C"
