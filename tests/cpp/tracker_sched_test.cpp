// tracker_sched_test.cpp -- CPU test of kcc_tracker.cpp's batching (look-ahead batches, keyframe-chain guesses, prefetched windows)
// against a STUB of the C ABI underneath it: every nik_* entry point the tracker calls is replaced by a host function whose
// "ComputePose" is a deterministic function of the two frames' contents, whose asynchronous batches deliver their results only
// when they are waited for (the result buffers are poisoned until then), and which checks what the real library's bookkeeping
// would enforce (slots written before they are read, batch sizes, result buffers alive).  The property: whatever the window
// size, look-ahead depth, batch room and prefetching, the tracker's outputs are bit-identical to pushing the frames one at a
// time -- MapBuilder::AddNewInput's per-frame loop (src/map_builder.cc:30-70, main.cpp:51-86).
//
// Build (tests/test_tracker_sched.py):  g++ -std=c++17 -O1 -ffp-contract=off tracker_sched_test.cpp ../../ni-slam_amd/csrc/kcc_tracker.cpp
#include "../../include/nislam_kcc.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <random>
#include <vector>

namespace {
const int FH = 4, FW = 8;                       // a "frame" is 32 bytes whose first four hold its content id
struct Truth { double x, y, a, psr_t, psr_r; };
std::vector<Truth> g_truth;                      // by content id
}

struct nik_ctx {
    int max_batch = 0, max_frames = 0;
    std::vector<int> content;                    // slot -> content id (-1: never written)
    struct Pending { nik_pose_result* res; std::vector<nik_pose_result> val; };
    std::deque<Pending> pending;
    long issued = 0, pairs = 0, early_reads = 0;
    long fail_at = -1;                           // nik_pose_batch_async number `fail_at` fails once (an injected device error)
};
struct nik_map { int unused; };

static nik_pose_result pose_of(const nik_ctx* c, nik_frame key, nik_frame cur) {
    nik_pose_result r; memset(&r, 0, sizeof(r));
    const int k = c->content[key], x = c->content[cur];
    if (k < 0 || x < 0) { fprintf(stderr, "stub: slot read before it was written\n"); abort(); }
    const Truth &K = g_truth[k], &X = g_truth[x];
    // relative motion in the key's frame, in pixels / radians (what ComputePose returns), rounded to the pixel grid the
    // arg-max lives on so that ties and exact repeats occur
    const double dx = X.x - K.x, dy = X.y - K.y, cs = std::cos(K.a), sn = std::sin(K.a);
    r.pose[0] = std::round(cs * dx + sn * dy); r.pose[1] = std::round(-sn * dx + cs * dy);
    r.pose[2] = std::round((X.a - K.a) * 720 / M_PI) * M_PI / 720;
    r.info[0] = r.info[1] = X.psr_t; r.info[2] = X.psr_r;
    return r;
}
static void deliver(nik_ctx* c, size_t upto) {
    for (size_t i = 0; i <= upto && !c->pending.empty(); ++i) {
        nik_ctx::Pending& P = c->pending.front();
        memcpy(P.res, P.val.data(), sizeof(nik_pose_result) * P.val.size());
        c->pending.pop_front();
    }
}

extern "C" {
int nik_get_dims(const nik_ctx* c, int d[6]) { d[0] = FH; d[1] = FW; d[2] = 0; d[3] = 0; d[4] = c->max_batch; d[5] = c->max_frames; return NIK_OK; }
int nik_set_lane_rotation(nik_ctx*, int) { return NIK_OK; }
int nik_set_call_depth(nik_ctx*, int) { return NIK_OK; }
int nik_synchronize(nik_ctx* c) { if (!c->pending.empty()) deliver(c, c->pending.size() - 1); return NIK_OK; }
int nik_intermedium_batch_dev(nik_ctx* c, int n, const uint8_t* d_gray, const nik_frame* dst) {
    if (n > c->max_batch) return NIK_ERR_CAPACITY;
    for (int i = 0; i < n; ++i) {
        if (dst[i] < 0 || dst[i] >= c->max_frames) return NIK_ERR_INVALID_ARG;
        int id; memcpy(&id, d_gray + (size_t)i * FH * FW, 4);
        c->content[dst[i]] = id;
    }
    return NIK_OK;
}
int nik_intermedium_u8(nik_ctx* c, const uint8_t* gray, int, nik_frame dst) { nik_synchronize(c); return nik_intermedium_batch_dev(c, 1, gray, &dst); }
int nik_pose(nik_ctx* c, nik_frame key, nik_frame cur, int, double pose[3], double info[3], nik_pose_result* res) {
    nik_synchronize(c);
    const nik_pose_result r = pose_of(c, key, cur);
    if (pose) memcpy(pose, r.pose, sizeof(r.pose));
    if (info) memcpy(info, r.info, sizeof(r.info));
    if (res) *res = r;
    return NIK_OK;
}
int nik_pose_batch_async(nik_ctx* c, int n, const nik_frame* keys, const nik_frame* curs, int, nik_pose_result* res) {
    if (n <= 0 || n > c->max_batch) return NIK_ERR_CAPACITY;
    if (c->issued == c->fail_at) { c->fail_at = -1; return NIK_ERR_HIP; }
    nik_ctx::Pending P; P.res = res; P.val.resize(n);
    for (int i = 0; i < n; ++i) P.val[i] = pose_of(c, keys[i], curs[i]);
    memset(res, 0xFF, sizeof(nik_pose_result) * (size_t)n);                  // not final until waited for
    c->pending.push_back(std::move(P)); c->issued += 1; c->pairs += n;
    return NIK_OK;
}
int nik_wait_results(nik_ctx* c, const nik_pose_result* res, int n) {
    int last = -1;
    for (size_t i = 0; i < c->pending.size(); ++i) if (c->pending[i].res >= res && c->pending[i].res < res + n) last = (int)i;
    if (last >= 0) deliver(c, (size_t)last);
    return NIK_OK;
}
int nik_pose_batch(nik_ctx* c, int n, const nik_frame* keys, const nik_frame* curs, int nlr, nik_pose_result* res) {
    const int rc = nik_pose_batch_async(c, n, keys, curs, nlr, res);
    return rc ? rc : nik_synchronize(c);
}
int nik_dev_malloc(nik_ctx*, size_t bytes, void** out) { *out = malloc(bytes); return *out ? NIK_OK : NIK_ERR_HIP; }
int nik_dev_free(nik_ctx*, void* p) { free(p); return NIK_OK; }
int nik_upload_u8_async(nik_ctx*, int n, const uint8_t* gray, int stride, size_t frame_stride, uint8_t* d_dst) {
    for (int i = 0; i < n; ++i) for (int y = 0; y < FH; ++y) memcpy(d_dst + ((size_t)i * FH + y) * FW, gray + (size_t)i * frame_stride + (size_t)y * stride, FW);
    return 1;
}
int nik_upload_fence(nik_ctx*, int) { return NIK_OK; }
int nik_upload_wait(nik_ctx*) { return NIK_OK; }
int nik_upload_after_compute(nik_ctx*) { return NIK_OK; }
int nik_map_add_frame(nik_map*, int32_t, nik_frame, const double*, const double*) { return NIK_OK; }
int nik_map_find_loop(nik_map*, int32_t, const double*, nik_loop_result* out) { memset(out, 0, sizeof(*out)); return NIK_OK; }
int nik_map_update_poses(nik_map*, int, const int32_t*, const double*) { return NIK_OK; }
int nik_pose_graph_optimize(int, const int32_t*, double*, int, const nik_pg_constraint*, int, nik_pg_summary* s) { if (s) memset(s, 0, sizeof(*s)); return NIK_OK; }
}

namespace {

nik_tracker_config config() {
    nik_tracker_config c; memset(&c, 0, sizeof(c));
    c.fx = c.fy = 600; c.cx = FW / 2 - 0.5; c.cy = FH / 2 + 0.25; c.height = 0.1;
    c.extrinsics[0] = c.extrinsics[4] = c.extrinsics[8] = 1;
    c.max_distance = 0.03; c.max_angle = 0.02; c.lower_response_thr = 8; c.upper_response_thr = 9;
    return c;
}

// a camera path whose keyframe gaps are regular for a while, then periodic, then irregular, with frames whose PSR falls
// between the thresholds (inserted by rule c3/c4) or below them (bad tracking)
std::vector<uint8_t> make_sequence(int n, unsigned seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> U(0, 1);
    g_truth.assign(n, Truth());
    double x = 0, y = 0, a = 0, vx = 3, vy = 1, va = 0.001;
    for (int i = 0; i < n; ++i) {
        const int phase = (i / 97) % 3;
        if (phase == 0) { vx = 3; vy = 1; va = 0.001; }                                       // steady: regular gaps
        else if (phase == 1) { vx = (i % 7 < 4) ? 5 : 1.5; vy = 0.5; va = 0.004; }            // periodic
        else if (U(rng) < 0.3) { vx = 8 * U(rng); vy = 4 * U(rng) - 2; va = 0.02 * U(rng); }  // irregular
        if (U(rng) < 0.02) { x += 40 * U(rng); }                                              // a jump
        x += vx; y += vy; a += va;
        const double p = U(rng);
        g_truth[i] = { x, y, a, p < 0.03 ? 8.5 : (p < 0.05 ? 5.0 : 20 + U(rng)), U(rng) < 0.02 ? 8.7 : 30.0 };
    }
    std::vector<uint8_t> frames((size_t)n * FH * FW, 0);
    for (int i = 0; i < n; ++i) memcpy(frames.data() + (size_t)i * FH * FW, &i, 4);
    return frames;
}

struct Run { std::vector<nik_track_output> out; long calls = 0, pairs = 0, held = 0, failed = 0; };

// mode 0: push_dev window by window; 1: with the next window prefetched; 2: with two windows prefetched; 3: push_host; 4: push_u8
Run run(const std::vector<uint8_t>& frames, int n, int window, int mode, int depth, int room, unsigned jitter_seed = 0) {
    char b[32];
    snprintf(b, sizeof b, "%d", depth); setenv("NIK_TRK_DEPTH", b, 1);
    snprintf(b, sizeof b, "%d", room); setenv("NIK_TRK_FLIGHT", b, 1);
    nik_ctx ctx; ctx.max_batch = window; ctx.max_frames = n + 3 * window + 2; ctx.content.assign(ctx.max_frames, -1);
    const nik_tracker_config cfg = config();
    nik_tracker* t = nullptr;
    if (nik_tracker_create(&ctx, &cfg, &t)) { fprintf(stderr, "create failed\n"); exit(2); }
    Run R; R.out.resize(n);
    const size_t fb = (size_t)FH * FW;
    std::mt19937 rng(jitter_seed);
    int rc = 0;
    if (mode == 3) rc = nik_tracker_push_host(t, n, frames.data(), FW, fb, R.out.data());
    else if (mode == 4) { for (int i = 0; i < n && !rc; ++i) rc = nik_tracker_push_u8(t, frames.data() + i * fb, FW, &R.out[i]); }
    else {
        // (jitter: windows of varying length, as a caller at the end of a file or with a variable camera rate would push them)
        std::vector<int> starts;
        for (int b0 = 0; b0 < n;) { starts.push_back(b0); b0 += jitter_seed ? 1 + (int)(rng() % window) : window; }
        starts.push_back(n);
        const int nw = (int)starts.size() - 1;
        int next_pre = 1, in_pre = 0;                               // next window to prefetch; windows the tracker holds as prefetched (at most two)
        for (int k = 0; k < nw && !rc; ++k) {
            next_pre = std::max(next_pre, k + 1);
            while (mode >= 1 && in_pre < 2 && next_pre < nw && next_pre <= k + mode && !rc) {
                rc = nik_tracker_prefetch_dev(t, starts[next_pre + 1] - starts[next_pre], frames.data() + starts[next_pre] * fb);
                next_pre += 1; in_pre += 1;
            }
            if (k > 0 && mode >= 1 && in_pre > 0) in_pre -= 1;          // (window k was prefetched: the push takes it over)
            if (!rc) rc = nik_tracker_push_dev(t, starts[k + 1] - starts[k], frames.data() + starts[k] * fb, R.out.data() + starts[k]);
        }
    }
    if (rc) { fprintf(stderr, "push failed: %d (mode %d window %d depth %d room %d)\n", rc, mode, window, depth, room); exit(2); }
    long sp[3]; nik_tracker_speculation(t, sp);
    R.held = sp[0]; R.failed = sp[1]; R.calls = ctx.issued; R.pairs = ctx.pairs;
    nik_tracker_destroy(t);
    if (!ctx.pending.empty()) { fprintf(stderr, "results left in flight after destroy\n"); exit(2); }
    return R;
}

bool same(const nik_track_output& a, const nik_track_output& b);

// an injected failure of batch number `fail_at`: the push returns the error, nothing stays in flight, the frames decided before
// it equal the reference's, the undecided ones are zero -- and pushing again from the first undecided frame carries on exactly
// as if nothing had happened (the tracker's state is only ever advanced by results it has applied, in frame order)
bool run_with_failure(const std::vector<uint8_t>& frames, int n, int window, int depth, long fail_at, const std::vector<nik_track_output>& ref) {
    char b[32];
    snprintf(b, sizeof b, "%d", depth); setenv("NIK_TRK_DEPTH", b, 1); setenv("NIK_TRK_FLIGHT", "0", 1);
    nik_ctx ctx; ctx.max_batch = window; ctx.max_frames = n + 3 * window + 2; ctx.content.assign(ctx.max_frames, -1); ctx.fail_at = fail_at;
    const nik_tracker_config cfg = config();
    nik_tracker* t = nullptr;
    if (nik_tracker_create(&ctx, &cfg, &t)) return false;
    std::vector<nik_track_output> out(n);
    const size_t fb = (size_t)FH * FW;
    bool ok = true, failed = false;
    for (int b0 = 0; b0 < n && ok;) {
        const int m = std::min(window, n - b0);
        const int rc = nik_tracker_push_dev(t, m, frames.data() + b0 * fb, out.data() + b0);
        int done = m;
        if (rc) {
            ok = ok && rc == NIK_ERR_HIP && !failed && ctx.pending.empty();
            failed = true;
            done = 0;
            while (done < m && (b0 + done == 0 ? out[0].inserted != 0 : out[b0 + done].frame_id == b0 + done)) ++done;
            for (int i = done; i < m; ++i) { nik_track_output z; memset(&z, 0, sizeof z); ok = ok && !memcmp(&out[b0 + i], &z, sizeof z); }
        }
        b0 += done;
        if (rc && done == 0 && m == 0) break;
    }
    for (int i = 0; i < n && ok; ++i) ok = same(out[i], ref[i]);
    nik_tracker_destroy(t);
    return ok && failed && ctx.pending.empty();
}

bool same(const nik_track_output& a, const nik_track_output& b) {
    return a.frame_id == b.frame_id && a.inserted == b.inserted && a.good_tracking == b.good_tracking && a.key_frame_id == b.key_frame_id &&
           !memcmp(a.response, b.response, sizeof a.response) && !memcmp(a.cf_pose, b.cf_pose, sizeof a.cf_pose) &&
           !memcmp(a.robot_pose, b.robot_pose, sizeof a.robot_pose) && a.distance == b.distance && a.optimized == b.optimized;
}

}  // namespace

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 700;
    int bad = 0, cases = 0;
    for (unsigned seed = 1; seed <= 3; ++seed) {
        const std::vector<uint8_t> frames = make_sequence(n, seed);
        const Run ref = run(frames, n, 1, 4, 1, 0);                     // one frame at a time through push_u8: no batching at all
        int nkey = 0, ngood = 0;
        for (const nik_track_output& o : ref.out) { nkey += o.inserted; ngood += o.good_tracking; }
        if (nkey < n / 20 || nkey > n - n / 20 || ngood < n / 2) { printf("FAIL seed %u: degenerate sequence (%d keyframes, %d good)\n", seed, nkey, ngood); return 1; }
        const Run one = run(frames, n, 1, 0, 2, 0);                     // push_dev with windows of one frame
        for (int i = 0; i < n; ++i) if (!same(one.out[i], ref.out[i])) { printf("FAIL seed %u: push_dev(1) differs from push_u8 at frame %d\n", seed, i); return 1; }
        const int windows[] = { 2, 5, 16, 32, 64 };
        for (int w : windows)
            for (int mode = 0; mode <= 3; ++mode)
                for (int depth = 1; depth <= 4; ++depth)
                    for (int room : { 0, 3, 8, 24 })
                        for (unsigned jit : { 0u, 7u }) {
                            if (mode == 3 && jit) continue;
                            if (room > w) continue;
                            const Run r = run(frames, n, w, mode, depth, room, jit ? jit + seed : 0);
                            cases += 1;
                            for (int i = 0; i < n; ++i)
                                if (!same(r.out[i], ref.out[i])) {
                                    printf("FAIL seed %u window %d mode %d depth %d room %d jitter %u: frame %d differs (inserted %d/%d key %d/%d)\n", seed, w, mode,
                                           depth, room, jit, i, r.out[i].inserted, ref.out[i].inserted, r.out[i].key_frame_id, ref.out[i].key_frame_id);
                                    bad += 1; break;
                                }
                            if (w == 64 && mode == 1 && room == 0 && !jit)
                                printf("seed %u window 64 prefetch depth %d: %d keyframes, %ld batched calls, %ld pairs registered for %d frames, guesses %ld held / %ld failed\n",
                                       seed, depth, nkey, r.calls, r.pairs, n, r.held, r.failed);
                        }
    }
    {
        const std::vector<uint8_t> frames = make_sequence(n, 5);
        const Run ref = run(frames, n, 1, 4, 1, 0);
        int inj = 0;
        for (int w : { 8, 32 }) for (int depth = 1; depth <= 3; ++depth) for (long at : { 0L, 1L, 2L, 5L, 17L, 40L }) {
            inj += 1;
            if (!run_with_failure(frames, n, w, depth, at, ref.out)) { printf("FAIL: injected failure of batch %ld (window %d depth %d) not survived\n", at, w, depth); bad += 1; }
        }
        printf("injected failures: %d cases\n", inj);
    }
    printf("%s: %d configurations, %d differ\n", bad ? "FAIL" : "OK", cases, bad);
    return bad ? 1 : 0;
}
