"""End-to-end system test of everything around the hot path, wired the way the reference's MapBuilder wires it
(src/map_builder.cc:30-70,140-156,168-271): raw frames -> undistortion -> tracker (key frames) -> key-frame map + loop
closure -> KCC / loop edges -> pose-graph optimisation -> updated poses -> map stitcher recompute.
The registrations of integer-pixel synthetic motion are exact, so odometry drift is injected into the KCC edges on
purpose; the loop edges found by the GPU path must pull the trajectory back towards the ground truth."""
import math

import numpy as np
import pytest

import synth
from kcc_helpers import SMALL, nik
from ref_tracker import compute_absolute_pose, compute_relative_pose, normalize_angle


@pytest.mark.gpu
def test_full_pipeline_closes_a_loop():
    import torch
    N = nik()
    geom = SMALL; H, W = geom["H"], geom["W"]
    f, height = 600.0 * W / 640, 0.1
    cv = synth.canvas(61, H, W)
    path = [(2 * i, 3 * i) for i in range(10)] + [(2 * i, 3 * i + 1) for i in range(8, -1, -1)]      # out, then back beside it
    frames = np.stack([synth.window(cv, H, W, dy, dx) for dy, dx in path])
    n = len(frames)
    cfg = N.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    flow = N.CorrelationFlow(cfg, H, W, max_batch=6, max_frames=n + 8)
    # identity undistortion maps: the raw-frame path (remap fused into the conversion) is exercised without changing
    # the pictures, so the synthetic integer-pixel motion stays exact
    newK = (f, W / 2, f, H / 2)
    cc, rr = np.meshgrid(np.arange(W), np.arange(H))
    m1 = np.stack([cc, rr], axis=-1).astype(np.int16); m2 = np.zeros((H, W), np.uint16)
    flow.set_undistort(m1, m2)
    tc = N.tracker_config(fx=newK[0], fy=newK[2], cx=newK[1], cy=newK[3], height=height,
                          max_distance=0.002 * 80 / W * (W / 80), max_angle=0.02, lower_response_thr=8.0, upper_response_thr=9.0)
    trk = N.Tracker(flow, tc)
    kmap = N.KeyframeMap(flow, N.loop_config(grid_scale=0.01, frame_gap_thr=4, distance_thr=0.004, position_response_thr=12.0, angle_response_thr=12.0))
    trk.attach_map(kmap, True)
    st = N.Stitcher(flow, 64)
    d = torch.from_numpy(frames).cuda(); torch.cuda.synchronize()
    outs = []
    for b in range(0, n, 6):
        outs += trk.push_dev(d[b:b + min(6, n - b)].data_ptr(), min(6, n - b))
    keys = [o for o in outs if o["inserted"]]
    loops = trk.loops()
    assert len(keys) >= 8 and len(loops) >= 2
    # ground truth robot poses from the synthetic motion: the window moved by (dy, dx) px -> camera (dx, dy)/f * height
    truth = {o["frame_id"]: np.array([height * path[o["frame_id"]][1] / f, height * path[o["frame_id"]][0] / f, 0.0]) for o in keys}
    for o in keys:                                                                   # exact registrations: no drift yet
        assert np.allclose(o["robot_pose"][:2], truth[o["frame_id"]][:2], atol=1e-9)

    def to_camera(p):                                                                # ConvertImagePlanePoseToCamera (camera.cc:160-176)
        return np.array([p[0] / newK[0], p[1] / newK[2], p[2]])

    def to_robot(c):                                                                 # ConvertCameraPoseToRobot (:197-211), identity extrinsics
        return np.array([height * c[0], height * c[1], c[2]])
    # ---- edges as MapBuilder builds them: KCC edges between consecutive key frames (AddCFEdge, :140-145), loop edges
    # from the matches (AddLoopEdges, :180-191); OptimizeMap converts edge._T with ConvertCameraPoseToRobot (:236-238)
    rng = np.random.default_rng(0)
    cons, I3 = [], np.eye(3)
    for a, b in zip(keys[:-1], keys[1:]):
        rel_cam = compute_relative_pose(to_camera(np.array(a["cf_pose"])), to_camera(np.array(b["cf_pose"])))
        rel = to_robot(rel_cam) + np.array([2e-4, -1.5e-4, 0.002]) + rng.normal(0, [5e-5, 5e-5, 5e-4])   # injected odometry drift
        cons.append((a["frame_id"], b["frame_id"], rel[0], rel[1], rel[2], I3))
    n_odo = len(cons)
    for l in loops:
        rel = to_robot(to_camera(np.array(l["relative_pose"])))
        cons.append((l["loop_frame_id"], l["cur_frame_id"], rel[0], rel[1], rel[2], I3))
    ids = [o["frame_id"] for o in keys]
    # dead reckoning through the drifting odometry edges = the initial guess handed to the optimiser
    guess = {ids[0]: np.array(keys[0]["robot_pose"])}
    for (a, b, x, y, yaw, _) in cons[:n_odo]:
        guess[b] = compute_absolute_pose(guess[a], np.array([x, y, yaw]))
    err0 = max(np.hypot(*(guess[i][:2] - truth[i][:2])) for i in ids)
    opt, sm = N.pose_graph_optimize(ids, [guess[i] for i in ids], cons)
    assert sm["termination"] == 0 and sm["final_cost"] < sm["initial_cost"]
    err1 = max(np.hypot(*(opt[k][:2] - truth[i][:2])) for k, i in enumerate(ids))
    assert err1 < 0.6 * err0, (err0, err1)                                           # the loop edges pull the drift back
    # ---- stitcher: insert the key frames at their optimised poses, then move them and recompute (RecomputeOccupancy)
    def image_pose(robot):                                                           # ConvertRobotPoseToImagePlane + ConvertPrincipalToCenter, this camera
        return (robot[0] / height * newK[0], robot[1] / height * newK[2], robot[2])
    und = torch.empty_like(d)
    flow.undistort_dev(d.data_ptr(), n, und.data_ptr())
    assert torch.equal(und, d)                                                       # identity maps
    for k, i in enumerate(ids):
        st.insert_dev(i, und[i].data_ptr(), image_pose(guess[i]))
    cells_before = {c: st.read_cell(*c)[1].sum() for c in st.cells()}
    st.recompute(ids, [image_pose(opt[k]) for k in range(len(ids))])
    cells_after = {c: st.read_cell(*c)[1].sum() for c in st.cells()}
    assert sum(cells_before.values()) == sum(cells_after.values()) == len(ids) * H * W        # every pixel of every key frame lands somewhere
    trk.close(); kmap.close(); st.close(); flow.close()


@pytest.mark.gpu
def test_tracker_runs_check_and_optimize_like_map_builder():
    """MapBuilder::CheckAndOptimize in the C++ tracker (map_builder.cc:66-67,108-116,180-277): loops found at consecutive key
    frames accumulate; the first key frame without a loop adds their edges, optimises the pose graph, rewrites the poses
    (Map::UpdatePoses, UpdateValueAfterLoop) and clears the matches."""
    import torch
    N = nik()
    geom = SMALL; H, W = geom["H"], geom["W"]
    f, height = 600.0 * W / 640, 0.1
    cv = synth.canvas(61, H, W)
    # out, back beside the outbound track (loops at consecutive key frames), then away into fresh ground (no loop)
    path = [(2 * i, 3 * i) for i in range(10)] + [(2 * i, 3 * i + 1) for i in range(8, -1, -1)] + [(-3 * i, -2 * i) for i in range(1, 7)]
    frames = np.stack([synth.window(cv, H, W, dy, dx) for dy, dx in path])
    n = len(frames)
    cfg = N.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    flow = N.CorrelationFlow(cfg, H, W, max_batch=6, max_frames=n + 8)
    tc = N.tracker_config(fx=f, fy=f, cx=W / 2, cy=H / 2, height=height, max_distance=0.002, max_angle=0.02, lower_response_thr=8.0, upper_response_thr=9.0)
    trk = N.Tracker(flow, tc)
    kmap = N.KeyframeMap(flow, N.loop_config(grid_scale=0.01, frame_gap_thr=4, distance_thr=0.004, position_response_thr=12.0, angle_response_thr=12.0))
    trk.attach_map(kmap, True)
    d = torch.from_numpy(frames).cuda(); torch.cuda.synchronize()
    outs = []
    for b in range(0, n, 6):
        outs += trk.push_dev(d[b:b + min(6, n - b)].data_ptr(), min(6, n - b))
    keys = [o for o in outs if o["inserted"]]
    loops = trk.loops()
    nopt, sm = trk.optimizations()
    opt_frames = [o["frame_id"] for o in outs if o["optimized"]]
    assert len(loops) >= 2 and nopt >= 1 and len(opt_frames) == nopt and sm["termination"] in (0, 1)
    assert trk.pending_loops() == 0 or keys[-1]["frame_id"] in [l["cur_frame_id"] for l in loops]
    edges = trk.edges()
    n_loop_edges = sum(1 for e in edges if e[6] == 1)
    assert sum(1 for e in edges if e[6] == 0) == len(keys) - 1            # one KCC edge per key frame after the first
    assert 2 <= n_loop_edges <= len(loops)                                # loops still pending at the end have no edge yet
    # every optimisation consumed at least two consecutive loops, none of them twice
    cur_ids = [l["cur_frame_id"] for l in loops]
    assert len(set(cur_ids)) == len(cur_ids)
    # the optimised poses are a fixed point of the solver on the edge set of the LAST optimisation (later key frames only
    # append dead-reckoned poses and KCC edges, which cost nothing)
    ids, poses = trk.poses()
    assert ids == [o["frame_id"] for o in keys]
    again, sm2 = N.pose_graph_optimize(ids, poses, [e[:6] for e in edges])
    assert np.abs(again - poses).max() < 1e-7 and sm2["final_cost"] <= sm2["initial_cost"] + 1e-18
    # exact registrations of integer-pixel motion: the optimum is the ground truth itself
    for k, o in enumerate(keys):
        dy, dx = path[o["frame_id"]]
        assert np.allclose(poses[k][:2], [height * dx / f, height * dy / f], atol=1e-8)
    trk.close(); kmap.close(); flow.close()
