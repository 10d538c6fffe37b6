"""CPU restatement of the reference's keyframe Map and LoopClosure candidate logic, on top of the CPU oracle.
TEST INFRASTRUCTURE ONLY (the checker for ni-slam_amd/csrc/kcc_map.cpp).

Follows /root/reference/src/map.cc:17-30 (AddFrame), :32-34,58-64 (frame distance), :81-101 (grid),
/root/reference/src/loop_closure.cc:10-73 (the three FindLoopClosure overloads), include/loop_closure.h:8-25.
Unspecified in the reference: iteration order of the per-cell unordered_set -- fixed here (and in the product) to
ascending frame id; it only matters for exact ties of response.sum()."""
import numpy as np


class RefMap:
    def __init__(self, grid_scale=0.1, frame_gap_thr=100, distance_thr=5.0, position_response_thr=60.0, angle_response_thr=60.0):
        self.grid_scale, self.frame_gap_thr, self.distance_thr = grid_scale, frame_gap_thr, distance_thr
        self.position_response_thr, self.angle_response_thr = position_response_thr, angle_response_thr
        self.frames = {}      # id -> dict(pose, distance or None, payload)
        self.grid = {}        # (gx, gy) -> [ids]

    def cell(self, x, y):                                         # map.cc:81-85: static_cast<int> truncates toward zero
        return (int(x / self.grid_scale), int(y / self.grid_scale))

    def add_frame(self, frame_id, pose, distance=None, payload=None):
        if not self.frames:
            frame_id = 0                                          # map.cc:18-21
        assert frame_id not in self.frames
        self.frames[frame_id] = dict(pose=tuple(pose), distance=distance, payload=payload)
        self.grid.setdefault(self.cell(pose[0], pose[1]), []).append(frame_id)
        return frame_id

    def frame_distance(self, fid):                                # map.cc:58-64
        d = self.frames[fid]["distance"]
        return -1.0 if d is None else d

    def candidates(self, cur_id, prior_pose=None):
        if prior_pose is None:
            pool = sorted(self.frames)                            # std::map order
        else:
            cx, cy = self.cell(prior_pose[0], prior_pose[1])
            pool = []
            for i in (-1, 0, 1):
                for j in (-1, 0, 1):
                    pool += self.grid.get((cx + i, cy + j), [])
            pool = sorted(pool)
        out = []
        for fid in pool:
            if self.frame_gap_thr > 0 and abs(cur_id - fid) < self.frame_gap_thr:
                continue
            if self.distance_thr > 0 and abs(self.frame_distance(cur_id) - self.frame_distance(fid)) < self.distance_thr:
                continue
            out.append(fid)
        return out

    def find_loop(self, cur_id, compute_pose, prior_pose=None):
        """compute_pose(candidate_id) -> (pose[3], response[3]): ComputePose(candidate, current, not_large_rotation=False)"""
        best = dict(found=False, response=[-1.0, -1.0, -1.0], loop_frame_id=-1, relative_pose=[0.0, 0.0, 0.0])
        cands = self.candidates(cur_id, prior_pose)
        for fid in cands:
            pose, resp = compute_pose(fid)
            if float(np.sum(resp)) > float(np.sum(best["response"])):     # strict >, loop_closure.cc:61
                best.update(response=[float(v) for v in resp], loop_frame_id=fid, relative_pose=[float(v) for v in pose])
        best["found"] = best["response"][0] > self.position_response_thr and best["response"][2] > self.angle_response_thr
        best["n_candidates"] = len(cands)
        return best
