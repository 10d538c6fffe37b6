// kcc_map.cpp -- host-side keyframe map and loop-closure candidate management: the parts of the reference's Map
// (src/map.cc:17-30,32-34,58-64,81-101; include/map.h:15-45) and LoopClosure (src/loop_closure.cc:17-73) that decide
// WHICH keyframes a new keyframe is registered against.  The registrations themselves go through nik_match (all
// candidates batched on the GPU); frames are device slots, so the reference's per-candidate Frame::GetFFTResult
// copies (loop_closure.cc:55-56) disappear.
//
// Candidate order: the reference iterates a std::map (all frames: ascending id) or an unordered_set per grid cell
// (unspecified order); the winner is chosen with a strict '>' so order only breaks exact ties.  Here candidates are
// always visited in ascending frame id -- the tie goes to the lowest id.
#include "../../include/nislam_kcc.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <unordered_map>
#include <vector>

namespace {

struct KeyFrame { int id; nik_frame slot; double pose[3]; double distance; bool has_distance; };

struct Cell {
    int x, y;
    bool operator==(const Cell& o) const { return x == o.x && y == o.y; }
};
struct CellHash { size_t operator()(const Cell& c) const { return std::hash<long long>()(((long long)c.x << 32) ^ (unsigned)c.y); } };

}  // namespace

struct nik_map {
    nik_ctx* ctx = nullptr;
    nik_loop_config cfg{};
    std::map<int, KeyFrame> frames;                                    // Map::_frames (+ _frame_distanses)
    std::unordered_map<Cell, std::vector<int>, CellHash> grid;         // Map::_grid_map: cell -> frame ids

    // Map::ComputeGridLocation (map.cc:81-85): truncation toward zero, as static_cast<int>
    Cell cell_of(double x, double y) const { return Cell{ (int)(x / cfg.grid_scale), (int)(y / cfg.grid_scale) }; }

    // the loop-closure filters (loop_closure.cc:43-53)
    bool filtered(const KeyFrame& cur, const KeyFrame& f) const {
        if (cfg.frame_gap_thr > 0 && std::abs(cur.id - f.id) < cfg.frame_gap_thr) return true;
        if (cfg.distance_thr > 0) {
            const double d1 = cur.has_distance ? cur.distance : -1, d2 = f.has_distance ? f.distance : -1;   // Map::GetFrameDistance (:58-64)
            if (std::fabs(d1 - d2) < cfg.distance_thr) return true;
        }
        return false;
    }
    // candidate frame ids for the current frame: all frames, or the 3 x 3 grid cells around prior_pose
    // (loop_closure.cc:10-33), minus the filtered ones; ascending id
    int candidates(int cur_id, const double* prior_pose, std::vector<int>& out) const {
        out.clear();
        const auto it = frames.find(cur_id);
        if (it == frames.end()) return NIK_ERR_INVALID_ARG;
        const KeyFrame& cur = it->second;
        std::vector<int> pool;
        if (!prior_pose) {
            for (const auto& kv : frames) pool.push_back(kv.first);
        } else {
            const Cell c = cell_of(prior_pose[0], prior_pose[1]);
            for (int i = -1; i <= 1; ++i)
                for (int j = -1; j <= 1; ++j) {
                    const auto g = grid.find(Cell{ c.x + i, c.y + j });
                    if (g != grid.end()) pool.insert(pool.end(), g->second.begin(), g->second.end());
                }
            std::sort(pool.begin(), pool.end());
        }
        for (int id : pool) if (!filtered(cur, frames.at(id))) out.push_back(id);
        return NIK_OK;
    }
};

extern "C" {

int nik_map_create(nik_ctx* ctx, const nik_loop_config* cfg, nik_map** out) {
    if (!cfg || !out || !(cfg->grid_scale > 0)) return NIK_ERR_INVALID_ARG;
    nik_map* m = new nik_map();
    m->ctx = ctx; m->cfg = *cfg;
    *out = m;
    return NIK_OK;
}

void nik_map_destroy(nik_map* m) { delete m; }

int nik_map_add_frame(nik_map* m, int frame_id, nik_frame slot, const double pose[3], const double* distance) {
    if (!m || !pose) return NIK_ERR_INVALID_ARG;
    if (m->frames.empty()) frame_id = 0;                               // Map::AddFrame: the base frame gets id 0 (map.cc:18-21)
    if (m->frames.count(frame_id)) return NIK_ERR_INVALID_ARG;
    KeyFrame f{ frame_id, slot, { pose[0], pose[1], pose[2] }, distance ? *distance : 0.0, distance != nullptr };
    m->frames[frame_id] = f;
    m->grid[m->cell_of(pose[0], pose[1])].push_back(frame_id);         // with the pose at insertion time (map.cc:26-29)
    return NIK_OK;
}

int nik_map_size(const nik_map* m) { return m ? (int)m->frames.size() : 0; }

// Map::UpdatePoses (map.cc:73-79): poses only -- the grid keeps the cells the frames were inserted into, as the reference's does
int nik_map_update_poses(nik_map* m, int n, const int32_t* frame_ids, const double* poses) {
    if (!m || n < 0 || (n > 0 && (!frame_ids || !poses))) return NIK_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i) {
        auto it = m->frames.find(frame_ids[i]);
        if (it != m->frames.end()) for (int k = 0; k < 3; ++k) it->second.pose[k] = poses[3 * i + k];
    }
    return NIK_OK;
}

int nik_map_candidates(const nik_map* m, int cur_frame_id, const double* prior_pose, int* frame_ids, int cap, int* n) {
    if (!m || !n) return NIK_ERR_INVALID_ARG;
    std::vector<int> ids;
    const int rc = m->candidates(cur_frame_id, prior_pose, ids);
    if (rc) return rc;
    *n = (int)ids.size();
    for (int i = 0; i < *n && i < cap && frame_ids; ++i) frame_ids[i] = ids[i];
    return NIK_OK;
}

int nik_map_find_loop(nik_map* m, int cur_frame_id, const double* prior_pose, nik_loop_result* out) {
    if (!m || !out) return NIK_ERR_INVALID_ARG;
    if (!m->ctx) return NIK_ERR_INVALID_ARG;
    std::vector<int> ids;
    int rc = m->candidates(cur_frame_id, prior_pose, ids);
    if (rc) return rc;
    // LoopClosureResult(): found = false, response = (-1, -1, -1)  (loop_closure.h:14)
    *out = nik_loop_result{};
    out->cur_frame_id = cur_frame_id; out->loop_frame_id = -1; out->loop_slot = -1; out->n_candidates = (int)ids.size();
    for (int k = 0; k < 3; ++k) out->response[k] = -1.0;
    if (ids.empty()) return NIK_OK;
    std::vector<nik_frame> slots(ids.size());
    for (size_t i = 0; i < ids.size(); ++i) slots[i] = m->frames.at(ids[i]).slot;
    // ComputePose(candidate, current, not_large_rotation = false) for every candidate; keep the largest
    // response.sum() (strict >)  (loop_closure.cc:55-65)
    int best = -1; nik_pose_result br{};
    if ((rc = nik_match(m->ctx, m->frames.at(cur_frame_id).slot, (int)slots.size(), slots.data(), &best, nullptr, &br))) return rc;
    if (best >= 0) {
        out->loop_frame_id = ids[best]; out->loop_slot = slots[best];
        for (int k = 0; k < 3; ++k) { out->response[k] = br.info[k]; out->relative_pose[k] = br.pose[k]; }
    }
    out->found = (out->response[0] > m->cfg.position_response_thr) && (out->response[2] > m->cfg.angle_response_thr);   // :68-71
    return NIK_OK;
}

}  // extern "C"
