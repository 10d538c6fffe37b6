// kcc_tracker.cpp -- host-side sequence driver: the tracking subset of the reference's MapBuilder
// (src/map_builder.cc:30-70,86-138,158-166), SE(2) chaining (src/utils.cc:134-152) and the camera pose
// conversions (src/camera.cc:148-211), in double precision exactly as the reference, on top of the C ABI.
// No GPU code here; every registration goes through nik_track_batch_dev / nik_pose_batch.
#include "../../include/nislam_kcc.h"
#include "kcc_tune.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <vector>

namespace {

struct V3 { double v[3]; double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } };

// ceres::optimization_2d::NormalizeAngle (include/optimization_2d/normalize_angle.h:41-47)
inline double normalize_angle(double a) { const double two_pi = 2.0 * M_PI; return a - two_pi * std::floor((a + M_PI) / two_pi); }

// ComputeAbsolutePose (src/utils.cc:144-152) with RotationMatrix2D (pose_graph_2d_error_term.h:43-51)
inline V3 compute_absolute_pose(const V3& p1, const V3& rel) {
    const double c = std::cos(p1[2]), s = std::sin(p1[2]);
    V3 r;
    r[0] = p1[0] + (c * rel[0] - s * rel[1]);
    r[1] = p1[1] + (s * rel[0] + c * rel[1]);
    r[2] = normalize_angle(p1[2] + rel[2]);
    return r;
}
// ComputeRelativePose (src/utils.cc:134-142)
inline V3 compute_relative_pose(const V3& p1, const V3& p2) {
    const double c = std::cos(p1[2]), s = std::sin(p1[2]);
    const double dx = p2[0] - p1[0], dy = p2[1] - p1[1];
    V3 r;
    r[0] = c * dx + s * dy;          // Rw1^T * d
    r[1] = -s * dx + c * dy;
    r[2] = normalize_angle(p2[2] - p1[2]);
    return r;
}

}  // namespace

struct nik_tracker {
    nik_ctx* ctx = nullptr;
    nik_tracker_config cfg{};
    int H = 0, W = 0, max_batch = 0, max_frames = 0;
    bool init = false;
    int frame_id = 0;
    double distance = 0;
    nik_frame key_slot = -1; int key_frame_id = -1;
    V3 last_cf_pose{}, last_cf_real_pose{}, last_pose{};
    std::vector<nik_frame> free_slots;
    std::vector<nik_frame> keyframes;
    int spec_depth = 8;                      // frames registered speculatively per batch (adapts to the keyframe spacing)
    int last_gap = 0;                        // frames between the last two keyframes (0: unknown yet)
    std::vector<int> gap_hist;               // the recent keyframe gaps, oldest first (what the next gaps are guessed from)
    double guess_rate = 0.5;                 // running share of keyframe guesses that held
    long spec_hits = 0, spec_misses = 0, gpu_calls = 0;      // diagnostics (nik_tracker_speculation)
    long pairs_issued = 0, pairs_used = 0, pairs_behind_miss = 0;   // registrations enqueued / consumed / still in flight when a guess failed (nik_tracker_stats)

    // The gap that followed the most recent earlier occurrence of the current context -- the last sixteen gaps, else the last
    // fifteen, ... else the last one; without any such occurrence, the last gap again.  Regular spacing and periodic patterns
    // (6, 3, 6, 3, ... with an irregular beat every few periods) are guessed right; anything else costs a few wasted
    // registrations.
    static int guess_next_gap(const std::vector<int>& h) {
        const int n = (int)h.size();
        if (n == 0) return 0;
        for (int order = 16; order >= 1; --order) {
            if (n <= order) continue;
            for (int j = n - 2; j >= order - 1; --j) {
                bool same = true;
                for (int k = 0; k < order; ++k) same = same && h[j - k] == h[n - 1 - k];
                if (same) return h[j + 1];
            }
        }
        return h[n - 1];
    }
    nik_map* map = nullptr;                  // optional (borrowed): keyframes are added to it and searched for loops
    int to_find_loop = 0;
    // MapBuilder::_loop_matches: loops found at CONSECUTIVE keyframes; a keyframe without a loop triggers CheckAndOptimize
    // (optimise when >= 2 have accumulated) and clears them (map_builder.cc:66-67,108-116)
    std::vector<nik_loop_result> loops;
    std::vector<nik_loop_result> all_loops;  // every loop ever found (diagnostics: nik_tracker_loops)
    uint8_t* d_up[3] = { nullptr, nullptr, nullptr };   // upload ring of nik_tracker_push_host: three windows of max_batch frames on the device
    struct Pre { const uint8_t* ptr; int n; std::vector<nik_frame> slots; };
    std::vector<Pre> pre;                     // nik_tracker_prefetch_dev: windows whose spectra are under way (at most two)
    // ---- look-ahead batches (nik_tracker_push_dev) ----
    // Frames are named by their frame id (gid: the id apply_result will give them -- frames are consumed in order, so the id of
    // every known frame is fixed the moment it is handed over).  A Flight is one asynchronous nik_pose_batch_async call: a cache
    // of ComputePose results for (key frame, current frame) pairs, planned from the tracker's GUESS of the coming key frames.  A
    // result is used iff its key is the frame the reference's rule really made the key frame and its slots are the frames'
    // slots -- so the outputs are those of frame-by-frame calls whatever was guessed; a wrong guess only wastes its GPU work.
    struct Seg { int key_gid; nik_frame key_slot; int first_gid, count, res_off; };
    struct Flight {
        std::vector<Seg> segs; std::vector<nik_frame> keys, curs; std::vector<nik_pose_result> res;
        int total = 0; bool waited = false;
    };
    std::deque<Flight> flights;               // in issue order; `waited` is prefix-closed
    int known0 = 0; std::vector<nik_frame> known;   // slot of frame gid known0 + i: the current push, then the prefetched windows
    // the planner's hypothetical state behind everything in flight: key frame, next frame to register, predicted next key frame
    struct Hyp { bool valid = false; int key_gid = -1; nik_frame key_slot = -1; int pos = 0, next_key = 0; std::vector<int> hist; } hyp;
    std::deque<int> pred;                     // predicted key frames (gids, ascending) not yet decided
    int plan_key_gid = -2;                    // key of the latest confirmed plan (a second one for the same key: its depth was too short)
    int la_depth = 2, la_room = 0;            // flights kept in flight; pairs per flight (0: max_batch)   $NIK_TRK_DEPTH / $NIK_TRK_FLIGHT

    // guessed keyframes planned ahead per batch: a wrong guess wastes what was queued behind it, so the chain is as long as the
    // guesses have been good -- about half the expected run of correct guesses, 1 / (1 - rate): 16 at 97 %, 5 at 90 %, 2 at 75 %
    int chain_cap() const { return std::max(1, std::min(16, (int)std::lround(0.5 / (1.0 - std::min(guess_rate, 0.97))))); }
    nik_frame slot_of(int gid) const { const int i = gid - known0; return (i >= 0 && i < (int)known.size()) ? known[i] : -1; }
    int known_end() const { return known0 + (int)known.size(); }
    int unwaited() const { int u = 0; for (const Flight& F : flights) u += !F.waited; return u; }
    // wait for flight j and everything issued before it
    int wait_flight(size_t j) {
        for (size_t i = 0; i <= j && i < flights.size(); ++i) {
            Flight& F = flights[i];
            if (F.waited) continue;
            const int rc = nik_wait_results(ctx, F.res.data(), F.total);
            if (rc) return rc;
            F.waited = true;
        }
        return NIK_OK;
    }
    int drop_flights() {
        int rc = flights.empty() ? NIK_OK : wait_flight(flights.size() - 1);
        flights.clear(); pred.clear(); hyp.valid = false;
        return rc;
    }
    // the flight / offset holding ComputePose(key, frame gid), or false
    bool find(int key_gid, nik_frame key_slot, int gid, size_t& fj, int& off) const {
        const nik_frame cs = slot_of(gid);
        for (size_t j = 0; j < flights.size(); ++j)
            for (const Seg& S : flights[j].segs)
                if (S.key_gid == key_gid && S.key_slot == key_slot && gid >= S.first_gid && gid < S.first_gid + S.count) {
                    const int o = S.res_off + (gid - S.first_gid);
                    if (flights[j].curs[o] == cs) { fj = j; off = o; return true; }
                }
        return false;
    }
    int issue(Flight& F) {
        F.total = (int)F.curs.size();
        if (F.total == 0) return NIK_OK;
        F.res.resize(F.total);
        flights.push_back(std::move(F));
        Flight& G = flights.back();
        gpu_calls += 1; pairs_issued += G.total;
        const int rc = nik_pose_batch_async(ctx, G.total, G.keys.data(), G.curs.data(), 1, G.res.data());
        if (rc) { (void)nik_wait_results(ctx, G.res.data(), G.total); flights.pop_back(); }      // (chunks already enqueued still write their results)
        return rc;
    }
    void add_seg(Flight& F, int key_gid, nik_frame key_slot, int first, int count) {
        F.segs.push_back({ key_gid, key_slot, first, count, (int)F.curs.size() });
        for (int i = 0; i < count; ++i) { F.keys.push_back(key_slot); F.curs.push_back(slot_of(first + i)); }
    }
    // continue the guessed chain from `hyp` into F: the frames up to and including the predicted next key frame against the
    // hypothetical key, then the frames behind that one against it, ... while F has room and frames are known
    void plan_chain(Flight& F, int room, int max_segs) {
        for (int c = 0; c < max_segs && hyp.valid; ++c) {
            const int count = std::min({ hyp.next_key - hyp.pos + 1, room - (int)F.curs.size(), known_end() - hyp.pos });
            if (count <= 0) break;
            add_seg(F, hyp.key_gid, hyp.key_slot, hyp.pos, count);
            hyp.pos += count;
            if (hyp.pos - 1 != hyp.next_key) break;                  // out of room or of frames: the segment goes on in the next flight
            hyp.hist.push_back(hyp.next_key - hyp.key_gid);
            pred.push_back(hyp.next_key);
            hyp.key_gid = hyp.next_key; hyp.key_slot = slot_of(hyp.key_gid);
            hyp.next_key = hyp.key_gid + std::max(1, guess_next_gap(hyp.hist));
        }
    }
    // Map::_edges (KCC edges between consecutive keyframes, loop edges) and the keyframes' robot poses
    struct EdgeRec { int from, to, type; double T[3]; };     // type 0 = KCC, 1 = Loop; T in camera units (edge->_T)
    std::vector<EdgeRec> edges;
    std::vector<int> kf_ids; std::vector<V3> kf_poses;       // Map::_frames: id -> pose (ascending id)
    int optimizations = 0; nik_pg_summary last_summary{};
    int map_rc = NIK_OK;                     // first error of a map / loop-closure / optimiser call inside a push

    // the map side of AddNewInput for a frame that became a keyframe (map_builder.cc:60-67,168-178): AddFrame,
    // SetFrameDistance, FindLoopClosure, and -- when this keyframe found no loop -- CheckAndOptimize.  Returns whether the
    // poses were optimised (the caller then refreshes its current pose: UpdateValueAfterLoop).
    bool keyframe_to_map(const nik_track_output& o, bool search) {
        kf_ids.push_back(o.frame_id);
        V3 rp0; for (int k = 0; k < 3; ++k) rp0[k] = o.robot_pose[k];
        kf_poses.push_back(rp0);
        if (!map || map_rc) return false;
        if ((map_rc = nik_map_add_frame(map, o.frame_id, o.slot, o.robot_pose, &o.distance))) return false;
        if (!search) return false;
        bool found = false;
        if (to_find_loop) {
            nik_loop_result lr;
            if ((map_rc = nik_map_find_loop(map, o.frame_id, o.robot_pose, &lr))) return false;     // prior = _current_pose (:169)
            if (lr.found) {
                V3 rp; for (int k = 0; k < 3; ++k) rp[k] = lr.relative_pose[k];
                rp = center_to_principal(rp);                                                  // :171
                for (int k = 0; k < 3; ++k) lr.relative_pose[k] = rp[k];
                loops.push_back(lr); all_loops.push_back(lr);
                found = true;
            }
        }
        return found ? false : check_and_optimize();
    }

    // MapBuilder::CheckAndOptimize (map_builder.cc:108-116): AddLoopEdges (:180-189), OptimizeMap (:195-271), Map::UpdatePoses
    bool check_and_optimize() {
        bool done = false;
        if (loops.size() >= 2) {
            for (const nik_loop_result& lm : loops) {                                           // AddLoopEdges
                V3 ip; for (int k = 0; k < 3; ++k) ip[k] = lm.relative_pose[k];
                const V3 cam = image_plane_to_camera(ip);
                edges.push_back({ lm.loop_frame_id, lm.cur_frame_id, 1, { cam[0], cam[1], cam[2] } });
            }
            // OptimizeMap: every frame's pose, every KCC / loop edge as a constraint in robot units, identity information
            std::vector<int32_t> ids(kf_ids.begin(), kf_ids.end());
            std::vector<double> poses(3 * kf_poses.size());
            for (size_t i = 0; i < kf_poses.size(); ++i) for (int k = 0; k < 3; ++k) poses[3 * i + k] = kf_poses[i][k];
            std::vector<nik_pg_constraint> cons;
            for (const EdgeRec& e : edges) {
                if (!std::binary_search(kf_ids.begin(), kf_ids.end(), e.from) || !std::binary_search(kf_ids.begin(), kf_ids.end(), e.to)) continue;
                V3 t; for (int k = 0; k < 3; ++k) t[k] = e.T[k];
                const V3 r = camera_to_robot(t);
                nik_pg_constraint c{};
                c.id_begin = e.from; c.id_end = e.to; c.x = r[0]; c.y = r[1]; c.yaw_radians = r[2];
                c.information[0] = c.information[4] = c.information[8] = 1.0;
                cons.push_back(c);
            }
            const int rc = nik_pose_graph_optimize((int)ids.size(), ids.data(), poses.data(), (int)cons.size(), cons.data(), 0, &last_summary);
            if (rc) { map_rc = rc; }
            // the solver also returns NIK_OK when its trust region collapsed (termination FAILURE): the reference CHECK-fails on an
            // unusable solution (OptimizeMap -> SolveOptimizationProblem, map_builder.cc:263-277); here the old poses are kept
            // and the error is reported through the push
            else if (last_summary.termination == NIK_PG_FAILURE) { map_rc = NIK_ERR_INVALID_ARG; }
            else {
                for (size_t i = 0; i < kf_poses.size(); ++i) for (int k = 0; k < 3; ++k) kf_poses[i][k] = poses[3 * i + k];
                if (map) map_rc = nik_map_update_poses(map, (int)ids.size(), ids.data(), poses.data());      // Map::UpdatePoses (map.cc:73-79)
                optimizations += 1; done = true;
            }
        }
        loops.clear();
        return done;
    }

    // Camera::ConvertRobotPoseToCamera (camera.cc:213-224) and ConvertCameraPoseToImagePlane (:178-195)
    V3 robot_to_camera(const V3& r) const {
        const double* E = cfg.extrinsics;
        const double det = E[0] * (E[4] * E[8] - E[5] * E[7]) - E[1] * (E[3] * E[8] - E[5] * E[6]) + E[2] * (E[3] * E[7] - E[4] * E[6]);
        const double inv[9] = { (E[4] * E[8] - E[5] * E[7]) / det, (E[2] * E[7] - E[1] * E[8]) / det, (E[1] * E[5] - E[2] * E[4]) / det,
                                (E[5] * E[6] - E[3] * E[8]) / det, (E[0] * E[8] - E[2] * E[6]) / det, (E[2] * E[3] - E[0] * E[5]) / det,
                                (E[3] * E[7] - E[4] * E[6]) / det, (E[1] * E[6] - E[0] * E[7]) / det, (E[0] * E[4] - E[1] * E[3]) / det };
        V3 c;
        for (int i = 0; i < 3; ++i) c[i] = inv[3 * i] * r[0] + inv[3 * i + 1] * r[1] + inv[3 * i + 2] * r[2];
        c[0] /= cfg.height; c[1] /= cfg.height;
        return c;
    }
    V3 camera_to_image_plane(const V3& c) const { V3 p; p[0] = cfg.fx * c[0]; p[1] = cfg.fy * c[1]; p[2] = c[2]; return p; }

    // Camera::ConvertCenterToPrincipal (src/camera.cc:148-158)
    V3 center_to_principal(const V3& c) const {
        const double cs = std::cos(c[2]), sn = std::sin(c[2]);
        const double bx = W * 0.5 - cfg.cx, by = H * 0.5 - cfg.cy;
        // (I - R) * O_bias, R = [[cs,-sn],[sn,cs]]
        V3 r;
        r[0] = c[0] + ((1 - cs) * bx + sn * by);
        r[1] = c[1] + (-sn * bx + (1 - cs) * by);
        r[2] = c[2];
        return r;
    }
    // Camera::ConvertImagePlanePoseToCamera (:160-176)
    V3 image_plane_to_camera(const V3& p) const { V3 r; r[0] = p[0] / cfg.fx; r[1] = p[1] / cfg.fy; r[2] = p[2]; return r; }
    // Camera::ConvertCameraPoseToRobot (:197-211): scale by the camera height, then the 3x3 extrinsics on (x, y, theta)
    V3 camera_to_robot(const V3& c) const {
        const double x = cfg.height * c[0], y = cfg.height * c[1], a = c[2];
        V3 r;
        for (int i = 0; i < 3; ++i) r[i] = cfg.extrinsics[3 * i] * x + cfg.extrinsics[3 * i + 1] * y + cfg.extrinsics[3 * i + 2] * a;
        return r;
    }
    V3 image_plane_to_robot(const V3& p) const { return camera_to_robot(image_plane_to_camera(p)); }
};

namespace {

// first frame: Initialize() (map_builder.cc:86-97): identity image-plane pose; it becomes the keyframe
void first_frame(nik_tracker* t, nik_frame slot, nik_track_output& o) {
    V3 cf{}; cf[0] = cf[1] = cf[2] = 0.0;
    const V3 real = t->image_plane_to_camera(cf), robot = t->camera_to_robot(real);
    memset(&o, 0, sizeof(o));
    o.frame_id = t->frame_id++; o.inserted = 1; o.good_tracking = 0; o.key_frame_id = -1; o.slot = slot;
    for (int k = 0; k < 3; ++k) { o.cf_pose[k] = cf[k]; o.robot_pose[k] = robot[k]; }
    t->distance = 0; t->init = true; o.distance = 0;
    t->last_cf_pose = cf; t->last_cf_real_pose = real; t->last_pose = robot;         // UpdateIntermedium (:99-106)
    t->key_slot = slot; t->key_frame_id = o.frame_id; t->keyframes.push_back(slot);
    t->keyframe_to_map(o, false);                                                       // Initialize(): AddFrame + distance 0, no search
}

// everything AddNewInput does with one ComputePose result (map_builder.cc:42-68); returns whether it was inserted
bool apply_result(nik_tracker* t, const nik_pose_result& r, nik_frame slot, nik_track_output& o) {
    memset(&o, 0, sizeof(o));
    o.frame_id = t->frame_id++; o.key_frame_id = t->key_frame_id; o.slot = -1;
    for (int k = 0; k < 3; ++k) o.response[k] = r.info[k];
    // Tracking() (:127-138)
    V3 rel; rel[0] = r.pose[0]; rel[1] = r.pose[1]; rel[2] = r.pose[2];
    rel = t->center_to_principal(rel);
    const bool good = r.info[0] > t->cfg.lower_response_thr && r.info[2] > t->cfg.lower_response_thr;
    o.good_tracking = good;
    V3 cur_cf = t->last_cf_pose, cur_real = t->last_cf_real_pose, cur_pose = t->last_pose;
    bool insert = false;
    if (good) {
        cur_cf = compute_absolute_pose(t->last_cf_pose, rel);
        cur_real = t->image_plane_to_camera(cur_cf);
        // UpdateCurrentPose() (:118-125)
        const V3 last_robot_cf = t->image_plane_to_robot(t->last_cf_pose), cur_robot_cf = t->image_plane_to_robot(cur_cf);
        cur_pose = compute_absolute_pose(t->last_pose, compute_relative_pose(last_robot_cf, cur_robot_cf));
        // ComputeRelativeDA() (:158-166) and the keyframe rule (:47-53)
        V3 d; for (int k = 0; k < 3; ++k) d[k] = cur_cf[k] - t->last_cf_pose[k];
        const V3 dc = t->image_plane_to_camera(d);
        const double dist = std::sqrt(dc[0] * dc[0] + dc[1] * dc[1]), ang = std::fabs(dc[2]);
        const bool c1 = dist > t->cfg.max_distance, c2 = ang > t->cfg.max_angle;
        const bool c3 = r.info[0] > t->cfg.lower_response_thr && r.info[0] < t->cfg.upper_response_thr;
        const bool c4 = r.info[2] > t->cfg.lower_response_thr && r.info[2] < t->cfg.upper_response_thr;
        insert = c1 || c2 || c3 || c4;
        if (insert) t->distance += dist;
    }
    for (int k = 0; k < 3; ++k) { o.cf_pose[k] = cur_cf[k]; o.robot_pose[k] = cur_pose[k]; }
    o.inserted = insert; o.distance = t->distance;
    if (insert) {
        // AddCFEdge() (:140-144): KCC edge last keyframe -> this frame, relative pose in camera units
        const V3 rel_real = compute_relative_pose(t->last_cf_real_pose, cur_real);
        t->edges.push_back({ t->key_frame_id, o.frame_id, 0, { rel_real[0], rel_real[1], rel_real[2] } });
        o.slot = slot;
        if (t->keyframe_to_map(o, true)) {
            // UpdateValueAfterLoop() (:273-277): the optimiser moved this frame
            cur_pose = t->kf_poses.back();
            cur_real = t->robot_to_camera(cur_pose);
            cur_cf = t->camera_to_image_plane(cur_real);
            for (int k = 0; k < 3; ++k) { o.cf_pose[k] = cur_cf[k]; o.robot_pose[k] = cur_pose[k]; }
            o.optimized = 1;
        }
        // UpdateIntermedium() (:99-106): this frame is the new keyframe
        t->last_cf_pose = cur_cf; t->last_cf_real_pose = cur_real; t->last_pose = cur_pose;
        t->key_slot = slot; t->key_frame_id = o.frame_id; t->keyframes.push_back(slot);
    }
    return insert;
}

}  // namespace

extern "C" {

int nik_tracker_create(nik_ctx* ctx, const nik_tracker_config* cfg, nik_tracker** out) {
    if (!ctx || !cfg || !out) return NIK_ERR_INVALID_ARG;
    int dims[6];
    int rc = nik_get_dims(ctx, dims);
    if (rc) return rc;
    if (cfg->fx == 0 || cfg->fy == 0 || cfg->height < 0) return NIK_ERR_INVALID_ARG;   // camera.cc:199-202 refuses height < 0
    nik_tracker* t = new nik_tracker();
    t->ctx = ctx; t->cfg = *cfg; t->H = dims[0]; t->W = dims[1]; t->max_batch = dims[4]; t->max_frames = dims[5];
    for (int s = t->max_frames - 1; s >= 0; --s) t->free_slots.push_back(s);          // pop_back hands out 0, 1, 2, ...
    if (const char* e = kcc::tune_env("NIK_TRK_DEPTH")) t->la_depth = std::max(1, std::min(4, atoi(e)));
    if (const char* e = kcc::tune_env("NIK_TRK_FLIGHT")) t->la_room = std::max(0, atoi(e));
    (void)nik_set_lane_rotation(ctx, 1);     // look-ahead batches run side by side on the context's streams ...
    (void)nik_set_call_depth(ctx, 4);        // ... and none of them makes the host wait for an older one when it is enqueued
    *out = t;
    return NIK_OK;
}

void nik_tracker_destroy(nik_tracker* t) {
    if (!t) return;
    (void)t->drop_flights();                 // (their result buffers are written when the calls retire)
    for (uint8_t* p : t->d_up) if (p) (void)nik_dev_free(t->ctx, p);
    delete t;
}

int nik_tracker_attach_map(nik_tracker* t, nik_map* m, int to_find_loop) {
    if (!t) return NIK_ERR_INVALID_ARG;
    if (t->init && m) return NIK_ERR_INVALID_ARG;         // the map must see every keyframe: attach before the first frame
    t->map = m; t->to_find_loop = to_find_loop;
    return NIK_OK;
}

int nik_tracker_loops(const nik_tracker* t, nik_loop_result* out, int cap, int* n) {
    if (!t || !n) return NIK_ERR_INVALID_ARG;
    *n = (int)t->all_loops.size();
    for (int i = 0; i < *n && i < cap && out; ++i) out[i] = t->all_loops[i];
    return NIK_OK;
}

int nik_tracker_pending_loops(const nik_tracker* t) { return t ? (int)t->loops.size() : 0; }

// diagnostics of the key-frame chain speculation of nik_tracker_push_dev: [guesses that held, guesses that failed, batched pose calls]
int nik_tracker_speculation(const nik_tracker* t, long out[3]) {
    if (!t || !out) return NIK_ERR_INVALID_ARG;
    out[0] = t->spec_hits; out[1] = t->spec_misses; out[2] = t->gpu_calls;
    return NIK_OK;
}

// more diagnostics: [guesses held, guesses failed, batches, registrations enqueued, registrations consumed, registrations of batches
// still in flight when a guess failed (their work is wasted), 0, 0]
int nik_tracker_stats(const nik_tracker* t, long out[8]) {
    if (!t || !out) return NIK_ERR_INVALID_ARG;
    const long v[8] = { t->spec_hits, t->spec_misses, t->gpu_calls, t->pairs_issued, t->pairs_used, t->pairs_behind_miss, 0, 0 };
    for (int i = 0; i < 8; ++i) out[i] = v[i];
    return NIK_OK;
}

// the tracker's guess of the next keyframe gap from a history of gaps (oldest first); exported for the tests
int nik_tracker_guess_gap(const int32_t* gaps, int n) {
    if (!gaps || n <= 0) return 0;
    return nik_tracker::guess_next_gap(std::vector<int>(gaps, gaps + n));
}

int nik_tracker_poses(const nik_tracker* t, int32_t* frame_ids, double* poses, int cap, int* n) {
    if (!t || !n) return NIK_ERR_INVALID_ARG;
    *n = (int)t->kf_ids.size();
    for (int i = 0; i < *n && i < cap; ++i) {
        if (frame_ids) frame_ids[i] = t->kf_ids[i];
        if (poses) for (int k = 0; k < 3; ++k) poses[3 * i + k] = t->kf_poses[i][k];
    }
    return NIK_OK;
}

int nik_tracker_edges(const nik_tracker* t, nik_pg_constraint* out, int32_t* types, int cap, int* n) {
    if (!t || !n) return NIK_ERR_INVALID_ARG;
    *n = (int)t->edges.size();
    for (int i = 0; i < *n && i < cap; ++i) {
        const auto& e = t->edges[i];
        V3 T; for (int k = 0; k < 3; ++k) T[k] = e.T[k];
        const V3 r = t->camera_to_robot(T);
        if (out) { nik_pg_constraint c{}; c.id_begin = e.from; c.id_end = e.to; c.x = r[0]; c.y = r[1]; c.yaw_radians = r[2];
                   c.information[0] = c.information[4] = c.information[8] = 1.0; out[i] = c; }
        if (types) types[i] = e.type;
    }
    return NIK_OK;
}

int nik_tracker_optimizations(const nik_tracker* t, nik_pg_summary* last) {
    if (!t) return 0;
    if (last) *last = t->last_summary;
    return t->optimizations;
}

int nik_tracker_keyframes(const nik_tracker* t, nik_frame* slots, int cap, int* n) {
    if (!t || !n) return NIK_ERR_INVALID_ARG;
    *n = (int)t->keyframes.size();
    for (int i = 0; i < *n && i < cap && slots; ++i) slots[i] = t->keyframes[i];
    return NIK_OK;
}

int nik_tracker_push_dev(nik_tracker* t, int n, const uint8_t* d_gray, nik_track_output* out) {
    if (!t || !d_gray || !out || n < 0) return NIK_ERR_INVALID_ARG;
    if (n == 0) return NIK_OK;
    if (n > t->max_batch) return NIK_ERR_CAPACITY;
    // a window whose spectra were started by nik_tracker_prefetch_dev (same pointer, same n) owns its slots already
    int pi = -1;
    for (size_t q = 0; q < t->pre.size(); ++q) if (t->pre[q].ptr == d_gray && t->pre[q].n == n) { pi = (int)q; break; }
    const bool pre = pi >= 0;
    if (!pre && (int)t->free_slots.size() < n) return NIK_ERR_CAPACITY;
    std::vector<nik_frame> slot(n);
    if (pre) { slot = t->pre[pi].slots; t->pre.erase(t->pre.begin() + pi); }
    else for (int i = 0; i < n; ++i) { slot[i] = t->free_slots.back(); t->free_slots.pop_back(); }
    int rc;
    memset(out, 0, sizeof(out[0]) * (size_t)n);
    // on an error: nothing stays in flight; frames that did not become keyframes give their slots back; outputs of unprocessed
    // frames stay zero
    auto bail = [&](int code) {
        (void)t->drop_flights();
        for (int i = n - 1; i >= 0; --i) if (!out[i].inserted) t->free_slots.push_back(slot[i]);
        return code;
    };
    // the spectra of a frame do not depend on the keyframe: all n frames in one batch
    if (!pre && (rc = nik_intermedium_batch_dev(t->ctx, n, d_gray, slot.data()))) return bail(rc);
    // the frames known from here on: this push, then the windows already prefetched (they will be pushed next, in that order)
    const int gid0 = t->frame_id, end_gid = gid0 + n;
    t->known0 = gid0; t->known = slot;
    for (const nik_tracker::Pre& P : t->pre) t->known.insert(t->known.end(), P.slots.begin(), P.slots.end());
    // batches still in flight from the previous push were planned on its view of the coming frames: they stay usable iff that
    // view has come true (the window pushed now is the one that had been prefetched first)
    {
        bool same = true;
        for (const nik_tracker::Flight& F : t->flights)
            for (const nik_tracker::Seg& S : F.segs)
                for (int i = 0; i < S.count; ++i) {
                    const int g = S.first_gid + i;
                    if (g >= gid0 && F.curs[S.res_off + i] != t->slot_of(g)) same = false;
                    if (S.key_gid >= gid0 && S.key_slot != t->slot_of(S.key_gid)) same = false;
                }
        if (t->hyp.valid && t->hyp.key_gid >= gid0 && t->hyp.key_slot != t->slot_of(t->hyp.key_gid)) same = false;
        if (!same && (rc = t->drop_flights())) return bail(rc);
    }
    if (!t->init) first_frame(t, slot[0], out[0]);
    const int room = std::min(t->max_batch, t->la_room > 0 ? t->la_room : t->max_batch);
    while (t->frame_id < end_gid) {
        const int x = t->frame_id;
        // batches that only hold frames already decided are of no further use
        while (!t->flights.empty()) {
            bool behind = true;
            for (const nik_tracker::Seg& S : t->flights.front().segs) behind = behind && S.first_gid + S.count <= x;
            if (!behind) break;
            if ((rc = t->wait_flight(0))) return bail(rc);
            t->flights.pop_front();
        }
        size_t fj; int off;
        if (!t->find(t->key_frame_id, t->key_slot, x, fj, off)) {
            // Nothing in flight registers frame x against the current keyframe: plan from the confirmed state.
            //
            // The first segment registers the next `depth` frames against the current keyframe.  The registrations are
            // speculative: they are valid up to and including the next inserted frame, the ones after it are redone against
            // the new keyframe.  The depth follows the observed keyframe spacing (twice the guessed gap), so little work is
            // thrown away while the batches stay as large as the sequence allows.
            //
            // Key-frame chains.  Every keyframe switch would otherwise cost one host round trip of a small, latency-bound batch
            // (~0.1 ms whatever its size below 16 pairs).  The gaps between keyframes are guessed from their own history
            // (guess_next_gap: regular spacing and short periodic patterns): the SAME batch also registers the frames behind the
            // guessed next keyframe against it (every frame's spectra are resident: any frame can serve as a key), and behind the
            // one guessed after that, and so on while the batch has room -- and the batches after it carry the chain on
            // (plan_chain) while this one runs.
            t->pred.clear(); t->hyp.valid = false;
            if (t->plan_key_gid == t->key_frame_id)        // the previous plan for this keyframe ended without an insertion: look further
                t->spec_depth = std::min(t->max_batch, 2 * std::max(1, t->spec_depth));
            t->plan_key_gid = t->key_frame_id;
            const int guess = nik_tracker::guess_next_gap(t->gap_hist);      // frames from the current keyframe to the next one (0: no history)
            // (never below the adaptive depth: it doubles after a plan without an insertion, so the window still grows towards
            // max_batch when the key frames suddenly come further apart than their history says)
            const int depth = std::max(1, guess > 0 ? std::max({ 4, 2 * guess, t->spec_depth }) : t->spec_depth);
            const int m0 = std::min({ t->known_end() - x, depth, room });
            nik_tracker::Flight F;
            t->add_seg(F, t->key_frame_id, t->key_slot, x, m0);
            // the first guessed keyframe must be one of the frames the first segment registers -- that is what confirms it
            const int kg = t->key_frame_id + guess;
            if (guess > 0 && kg >= x && kg < x + m0 && kg < t->known_end() - 1) {
                t->hyp.valid = true; t->hyp.hist = t->gap_hist;
                t->hyp.hist.push_back(guess); t->pred.push_back(kg);
                t->hyp.key_gid = kg; t->hyp.key_slot = t->slot_of(kg); t->hyp.pos = kg + 1;
                t->hyp.next_key = kg + std::max(1, nik_tracker::guess_next_gap(t->hyp.hist));
                // a guess costs its few registrations, a round trip saved is worth ~25 of them: keep guessing as long as one
                // guess in four holds; otherwise a single probe per plan keeps the rate measured
                t->plan_chain(F, room, t->chain_cap());
                if (t->guess_rate < 0.25) t->hyp.valid = false;
            }
            if ((rc = t->issue(F))) return bail(rc);
            continue;
        }
        if (!t->flights[fj].waited) {
            // about to block on a batch: first hand the GPU the batches behind it (the chain carried on from the planner's state)
            while (t->hyp.valid && t->unwaited() < t->la_depth && t->hyp.pos < t->known_end()) {
                nik_tracker::Flight F;
                t->plan_chain(F, room, t->chain_cap());
                if (F.curs.empty()) break;
                if ((rc = t->issue(F))) return bail(rc);
            }
            if ((rc = t->wait_flight(fj))) return bail(rc);
        }
        const int prev_key_frame = t->key_frame_id;
        const nik_pose_result r = t->flights[fj].res[off];
        t->pairs_used += 1;
        const bool inserted = apply_result(t, r, slot[x - gid0], out[x - gid0]);
        if (inserted) {
            t->last_gap = std::max(1, t->key_frame_id - prev_key_frame);
            t->gap_hist.push_back(t->last_gap);
            if (t->gap_hist.size() > 96) t->gap_hist.erase(t->gap_hist.begin());
            t->spec_depth = std::min(t->max_batch, std::max(4, 2 * t->last_gap));
        }
        // the guesses: a predicted keyframe must be inserted, and nothing before it
        if (!t->pred.empty() && (inserted || t->pred.front() == x)) {
            if (inserted && t->pred.front() == x) { t->pred.pop_front(); t->spec_hits += 1; t->guess_rate = 0.9 * t->guess_rate + 0.1; }
            else {
                t->pred.clear(); t->hyp.valid = false; t->spec_misses += 1; t->guess_rate = 0.9 * t->guess_rate;
                for (const nik_tracker::Flight& F : t->flights) if (!F.waited) t->pairs_behind_miss += F.total;
            }
        } else if (inserted) {
            t->hyp.valid = false;                       // (an insertion nobody predicted: whatever the planner assumed is off)
        }
    }
    // recycle the slots of frames that did not become keyframes
    for (int i = n - 1; i >= 0; --i) if (!out[i].inserted) t->free_slots.push_back(slot[i]);
    return t->map_rc;
}

int nik_tracker_push_u8(nik_tracker* t, const uint8_t* gray, int stride, nik_track_output* out) {
    if (!t || !gray || !out) return NIK_ERR_INVALID_ARG;
    if (t->free_slots.empty()) return NIK_ERR_CAPACITY;
    { const int rcf = t->drop_flights(); if (rcf) return rcf; }      // (look-ahead batches belong to the batched entry point)
    // host frame: upload through the single-frame entry point, then run the same logic with the spectra in place
    const nik_frame s = t->free_slots.back();
    int rc = nik_intermedium_u8(t->ctx, gray, stride, s);
    if (rc) return rc;
    t->free_slots.pop_back();
    if (!t->init) { first_frame(t, s, *out); return t->map_rc; }
    nik_pose_result r;
    if ((rc = nik_pose(t->ctx, t->key_slot, s, 1, nullptr, nullptr, &r))) { t->free_slots.push_back(s); return rc; }
    if (!apply_result(t, r, s, *out)) t->free_slots.push_back(s);
    return t->map_rc;
}

// ComputeIntermedium of the NEXT window, started now: the spectra of a frame do not depend on the key frame (map_builder.cc:33 runs
// ComputeFFTResult before anything else), so a streamed caller starts them for window k+1 before it pushes window k -- they
// run beside window k's registrations and their host round trips.  A later nik_tracker_push_dev with the same pointer and n
// picks them up (at most two windows may be under way); outputs are unchanged.
int nik_tracker_prefetch_dev(nik_tracker* t, int n, const uint8_t* d_gray) {
    if (!t || !d_gray || n <= 0 || t->pre.size() >= 2) return NIK_ERR_INVALID_ARG;
    if (n > t->max_batch || (int)t->free_slots.size() < n) return NIK_ERR_CAPACITY;
    nik_tracker::Pre P{ d_gray, n, std::vector<nik_frame>(n) };
    for (int i = 0; i < n; ++i) { P.slots[i] = t->free_slots.back(); t->free_slots.pop_back(); }
    const int rc = nik_intermedium_batch_dev(t->ctx, n, d_gray, P.slots.data());
    if (rc) { for (int i = n - 1; i >= 0; --i) t->free_slots.push_back(P.slots[i]); return rc; }
    t->pre.push_back(std::move(P));
    return NIK_OK;
}

// n host frames (cv::Mat CV_8UC1 each: H rows of `stride` bytes, `frame_stride` bytes apart) -- the reference's per-frame loop
// (main.cpp:51-86: GetImage -> AddNewInput) for a streamed caller.  The frames travel in windows of max_batch: while window k is
// registered, window k+1's spectra are computed (nik_tracker_prefetch_dev) and window k+2 is uploaded on the context's upload
// stream (nik_upload_u8_async: pinned sources by DMA, pageable ones through pinned staging), so neither the PCIe transfer nor
// ComputeIntermedium waits for the registrations' host round trips; outputs are those of n push_u8 calls.
int nik_tracker_push_host(nik_tracker* t, int n, const uint8_t* gray, int stride, size_t frame_stride, nik_track_output* out) {
    if (!t || !gray || !out || n < 0 || !t->pre.empty()) return NIK_ERR_INVALID_ARG;
    if (n == 0) return NIK_OK;
    const size_t fb = (size_t)t->H * t->W;
    const int win = t->max_batch, nw = (n + win - 1) / win;
    int rc;
    for (uint8_t*& p : t->d_up)
        if (!p) { void* q = nullptr; if ((rc = nik_dev_malloc(t->ctx, fb * win, &q))) return rc; p = (uint8_t*)q; }
    auto count = [&](int k) { return std::min(win, n - k * win); };
    auto upload = [&](int k) { return nik_upload_u8_async(t->ctx, count(k), gray + (size_t)k * win * frame_stride, stride, frame_stride, t->d_up[k % 3]); };
    std::vector<int> tk(nw + 2, -1);
    // on an error: no prefetched window stays behind (its slots go back, flights that name them are waited for and dropped, and
    // the `!pre.empty()` guard above does not lock the tracker out for good), and no upload of this call is still in flight
    auto bail = [&](int code) {
        (void)t->drop_flights();
        for (auto it = t->pre.rbegin(); it != t->pre.rend(); ++it)
            for (int i = it->n - 1; i >= 0; --i) t->free_slots.push_back(it->slots[i]);
        t->pre.clear();
        t->known.resize(std::min<size_t>(t->known.size(), (size_t)std::max(0, t->frame_id - t->known0)));
        (void)nik_upload_wait(t->ctx);
        return code;
    };
    if ((tk[0] = upload(0)) < 0) return bail(tk[0]);
    if (nw > 1 && (tk[1] = upload(1)) < 0) return bail(tk[1]);
    if ((rc = nik_upload_fence(t->ctx, tk[0]))) return bail(rc);
    // window 0 is prefetched like the others, so that the marker "the upload ring's buffer 0 has been read" sits right behind
    // its ComputeIntermedium batch -- not behind the whole of push(0) with its look-ahead pose batches, which stalled the staged
    // uploads of a pageable source
#ifndef KCC_TRK_OLD_MARKER
    if ((rc = nik_tracker_prefetch_dev(t, count(0), t->d_up[0]))) return bail(rc);
    if (nw > 3 && (rc = nik_upload_after_compute(t->ctx))) return bail(rc);
#endif
    for (int k = 0; k < nw; ++k) {
        if (k + 1 < nw) {
            // window k+1: its upload was enqueued a whole window ago -- wait for it on the device, start its spectra; window k+2:
            // start its upload (into the buffer of window k-1, whose push has returned)
            if ((rc = nik_upload_fence(t->ctx, tk[k + 1]))) return bail(rc);
            if (k + 2 < nw && (tk[k + 2] = upload(k + 2)) < 0) return bail(tk[k + 2]);
            if ((rc = nik_tracker_prefetch_dev(t, count(k + 1), t->d_up[(k + 1) % 3]))) return bail(rc);
            // The upload ring reuses a window's buffer three windows later; its last READER is that window's ComputeIntermedium
            // batch on the compute lanes (just enqueued for window k+1).  Uploads enqueued from here on wait for it on the device.
            // (Placed HERE, before this iteration's push: a marker behind the push would also cover its look-ahead pose batches,
            // and the staged uploads of a pageable source would stall the calling thread behind them -- measured 58 k -> 40 k
            // frames/s.  The intermedium batches are long done when the next upload is enqueued: no wait in practice.)
            static const bool no_order = kcc::tune_env("NIK_TRK_NO_UPLOAD_ORDER") != nullptr;      // (A/B switch of tools)
            if (!no_order && k + 3 < nw && (rc = nik_upload_after_compute(t->ctx))) return bail(rc);
        }
        if ((rc = nik_tracker_push_dev(t, count(k), t->d_up[k % 3], out + (size_t)k * win))) return bail(rc);
#ifdef KCC_TRK_OLD_MARKER                                 // (round-5 form, kept for the A/B of profiles/r06_push_host_marker.txt)
        if (k == 0 && nw > 3 && (rc = nik_upload_after_compute(t->ctx))) return bail(rc);
#endif
    }
    return nik_upload_wait(t->ctx);
}

}  // extern "C"
