"""Sequence driver: the C++ tracker over the HIP path against the CPU restatement of the
MapBuilder tracking subset over the oracle, on a seeded synthetic trajectory with keyframe switches."""
import math

import numpy as np
import pytest

import synth
from kcc_helpers import SMALL, FULL, nik
from oracle import kcc_oracle as ko
from ref_tracker import RefTracker, compute_absolute_pose, compute_relative_pose, normalize_angle


def _trajectory(geom, n, seed):
    """window drifting over one canvas: steady motion + slow rotation, so keyframes get inserted by the rule"""
    H, W = geom["H"], geom["W"]
    cv = synth.canvas(seed, H, W)
    rng = np.random.default_rng(seed)
    frames, dy, dx, th = [], 0.0, 0.0, 0.0
    step = max(1.0, min(H, W) / 60.0)
    for i in range(n):
        frames.append(synth.window(cv, H, W, int(round(dy)), int(round(dx)), round(th * 2) / 2))
        dy += step * rng.uniform(0.2, 1.0)
        dx += step * rng.uniform(-1.0, 0.6)
        th += rng.uniform(-0.4, 0.6)
    return np.stack(frames)


def test_se2_helpers():
    p1, p2 = np.array([1.0, 2.0, 0.3]), np.array([-0.5, 4.0, -2.9])
    rel = compute_relative_pose(p1, p2)
    back = compute_absolute_pose(p1, rel)
    assert np.allclose(back[:2], p2[:2]) and abs(normalize_angle(back[2] - p2[2])) < 1e-12
    assert normalize_angle(math.pi) == pytest.approx(-math.pi)


def test_keyframe_gap_guess():
    """the history-based guess behind push_dev's keyframe chains (host code, no GPU): regular spacing, periodic patterns with an
    irregular beat (bench.py's saw-tooth path: 6, 3, ... with 3, 3, 6, 1 every 64 frames), and the fallbacks"""
    N = nik()
    g = N.tracker_guess_gap
    assert g([]) == 0 and g([5]) == 5 and g([4, 4, 4]) == 4
    assert g([6, 3, 6, 3, 6]) == 3 and g([6, 3, 6, 3]) == 6
    assert g([2, 7]) == 7                                     # nothing to match: the last gap again
    period = [6, 3] * 7 + [3, 6, 1]                           # 64 frames
    hist = period * 4
    right = [g(hist[:k]) == hist[k] for k in range(len(hist))]
    assert all(right[2 * len(period):])                       # two periods of history: every later gap is guessed right
    assert sum(right[len(period):2 * len(period)]) >= len(period) // 2     # one period: the places behind the irregular beat are missed
    # a change of rhythm is picked up after one occurrence
    assert g([4, 4, 4, 9, 4, 4, 4]) == 9


@pytest.mark.gpu
@pytest.mark.parametrize("geom,n,window", [pytest.param(SMALL, 24, 8, id="60x80"), pytest.param(FULL, 10, 5, id="480x640")])
def test_tracker_matches_reference_logic(geom, n, window):
    import torch
    N = nik()
    H, W = geom["H"], geom["W"]
    frames = _trajectory(geom, n, 77)
    cfg = N.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    flow = N.CorrelationFlow(cfg, H, W, max_batch=window, max_frames=n + window + 2)
    # thresholds scaled to the geometry: small images have PSR ~ 20-30
    lower, upper = (12.0, 26.0) if geom is SMALL else (30.0, 140.0)
    tc = N.tracker_config(fx=600.0 * W / 640, fy=600.0 * W / 640, cx=W / 2 - 3.5, cy=H / 2 + 2.25, height=0.1,
                          max_distance=0.02, max_angle=0.02, lower_response_thr=lower, upper_response_thr=upper)
    trk = N.Tracker(flow, tc)
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    got = []
    for b in range(0, n, window):
        m = min(window, n - b)
        got += trk.push_dev(d[b:b + m].data_ptr(), m)
    ocfg = ko.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    ref = RefTracker(ko.Oracle(ocfg, H, W), H, W, fx=tc.fx, fy=tc.fy, cx=tc.cx, cy=tc.cy, height=tc.height,
                     max_distance=tc.max_distance, max_angle=tc.max_angle, lower=lower, upper=upper)
    want = [ref.add_new_input(f) for f in frames]
    n_key = 0
    for g, w in zip(got, want):
        assert (g["frame_id"], g["inserted"], g["good_tracking"], g["key_frame_id"]) == \
               (w["frame_id"], w["inserted"], w["good_tracking"], w["key_frame_id"]), (g, w)
        assert g["response"] == pytest.approx(w["response"], rel=3e-3, abs=1e-9)
        assert g["cf_pose"][:2] == pytest.approx(w["cf_pose"][:2], abs=1e-9)
        assert abs(normalize_angle(g["cf_pose"][2] - w["cf_pose"][2])) < 1e-9
        assert g["robot_pose"][:2] == pytest.approx(w["robot_pose"][:2], abs=1e-12)
        n_key += g["inserted"]
    assert 2 <= n_key < n, "the trajectory should exercise both keyframe insertion and plain tracking (%d)" % n_key
    assert len(trk.keyframes()) == n_key
    # the single-frame host entry point gives the same answers as the batched speculative one
    flow2 = N.CorrelationFlow(cfg, H, W, max_batch=2, max_frames=n + 2)
    trk2 = N.Tracker(flow2, tc)
    for f, g in zip(frames[: min(n, 8)], got):
        o = trk2.push_u8(f)
        for k in ("frame_id", "inserted", "good_tracking", "key_frame_id", "response", "cf_pose", "robot_pose"):
            assert o[k] == g[k], k
    trk.close(); trk2.close(); flow.close(); flow2.close()


@pytest.mark.gpu
def test_tracker_with_map_finds_loops():
    """MapBuilder::AddNewInput's map side (map_builder.cc:61-65,168-178): keyframes go into the map with pose and
    travelled distance and are searched for loops; an out-and-back trajectory must close on its early keyframes,
    with exactly the matches the restated reference logic (RefTracker + RefMap over the oracle) finds."""
    import torch
    from ref_map import RefMap
    N = nik()
    geom = SMALL; H, W = geom["H"], geom["W"]
    cv = synth.canvas(31, H, W)
    path = [(2 * i, 3 * i) for i in range(10)] + [(2 * i, 3 * i + 1) for i in range(8, -1, -1)]      # out, then back beside it
    frames = np.stack([synth.window(cv, H, W, dy, dx) for dy, dx in path])
    n = len(frames)
    cfg = N.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    flow = N.CorrelationFlow(cfg, H, W, max_batch=6, max_frames=n + 8)
    tc = N.tracker_config(fx=600.0 * W / 640, fy=600.0 * W / 640, cx=W / 2 - 1.5, cy=H / 2 + 0.75, height=0.1,
                          max_distance=0.002, max_angle=0.02, lower_response_thr=8.0, upper_response_thr=9.0)
    lkw = dict(grid_scale=0.01, frame_gap_thr=4, distance_thr=0.004, position_response_thr=12.0, angle_response_thr=12.0)
    trk = N.Tracker(flow, tc)
    kmap = N.KeyframeMap(flow, N.loop_config(**lkw))
    trk.attach_map(kmap, True)
    d = torch.from_numpy(frames).cuda(); torch.cuda.synchronize()
    got = []
    for b in range(0, n, 6):
        m = min(6, n - b)
        got += trk.push_dev(d[b:b + m].data_ptr(), m)
    loops = trk.loops()
    # ---- the same with the restated reference logic over the oracle
    ocfg = ko.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    ora = ko.Oracle(ocfg, H, W)
    ref = RefTracker(ora, H, W, fx=tc.fx, fy=tc.fy, cx=tc.cx, cy=tc.cy, height=tc.height,
                     max_distance=tc.max_distance, max_angle=tc.max_angle, lower=8.0, upper=9.0)
    rmap = RefMap(**lkw)
    spectra, want_loops = {}, []
    for i, f in enumerate(frames):
        w = ref.add_new_input(f)
        assert (got[i]["inserted"], got[i]["key_frame_id"]) == (w["inserted"], w["key_frame_id"]), i
        if not w["inserted"]:
            continue
        assert got[i]["distance"] == pytest.approx(ref.distance, abs=1e-12)
        f32 = ora.normalize_u8(f); spectra[i] = (f32,) + tuple(ora.intermedium(f32))
        rmap.add_frame(i, w["robot_pose"], ref.distance)
        if i == 0:
            continue

        def cp(fid, i=i):
            pose, info, _ = ora.compute_pose(spectra[fid][1], spectra[i][0], spectra[fid][2], spectra[i][2], False)
            return pose, info
        r = rmap.find_loop(i, cp, w["robot_pose"])
        if r["found"]:
            r["relative_pose"] = list(ref.center_to_principal(np.array(r["relative_pose"])))
            r["cur_frame_id"] = i
            want_loops.append(r)
    assert len(kmap) == len(rmap.frames) == sum(g["inserted"] for g in got)
    assert [(l["cur_frame_id"], l["loop_frame_id"]) for l in loops] == [(l["cur_frame_id"], l["loop_frame_id"]) for l in want_loops]
    assert len(loops) >= 2, "the way back must close loops on the way out"
    for g, w in zip(loops, want_loops):
        assert g["relative_pose"][:2] == pytest.approx(w["relative_pose"][:2], abs=1e-9)
        assert abs(normalize_angle(g["relative_pose"][2] - w["relative_pose"][2])) < 1e-6
        assert g["response"] == pytest.approx(w["response"], rel=5e-3)
    with pytest.raises(N.NikError):
        trk.attach_map(kmap, True)              # too late: the tracker has started
    trk.close(); kmap.close(); flow.close()


@pytest.mark.gpu
def test_keyframe_chain_speculation_changes_nothing():
    """A regular trajectory (a keyframe every few frames): push_dev registers the frames behind its GUESSED next keyframes in
    the batch that serves the current one.  The outputs must be exactly those of pushing the frames one at a time (no
    speculation possible there), most guesses must hold, and the look-ahead batches stay within one per keyframe and window."""
    import torch
    N = nik()
    geom = SMALL
    H, W = geom["H"], geom["W"]
    n, window = 96, 32
    cv = synth.canvas(31, H, W)
    frames = np.stack([synth.window(cv, H, W, (i % 24) - 12, 2 * (i % 24) - 24, 0.0) for i in range(n)])     # steady ramps of 24 frames
    cfg = N.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    tc = N.tracker_config(fx=600.0 * W / 640, fy=600.0 * W / 640, cx=W / 2 - 3.5, cy=H / 2 + 2.25, height=0.1,
                          max_distance=0.1, max_angle=0.02, lower_response_thr=8.0, upper_response_thr=9.0)
    d = torch.from_numpy(frames).cuda(); torch.cuda.synchronize()
    flow = N.CorrelationFlow(cfg, H, W, max_batch=window, max_frames=n + window + 2)
    trk = N.Tracker(flow, tc)
    got = []
    for b in range(0, n, window):
        got += trk.push_dev(d[b:b + window].data_ptr(), min(window, n - b))
    held, failed, calls = trk.speculation()
    flow1 = N.CorrelationFlow(cfg, H, W, max_batch=1, max_frames=n + 3)
    trk1 = N.Tracker(flow1, tc)
    one = []
    for i in range(n):
        one += trk1.push_dev(d[i:i + 1].data_ptr(), 1)
    for a, b in zip(got, one):
        for k in ("frame_id", "inserted", "good_tracking", "key_frame_id", "response", "cf_pose", "robot_pose", "distance"):
            assert a[k] == b[k], (a["frame_id"], k, a[k], b[k])
    n_key = sum(o["inserted"] for o in got)
    assert 8 <= n_key < n - 8, n_key
    assert held >= 4 and held > failed, (held, failed)
    # the batches are asynchronous look-ahead batches now: their number is no longer the number of host round trips.
    # While the guesses are young the chains are short (chain_cap follows the measured guess rate), so a 96-frame sequence may see
    # about one batch per keyframe; what must hold is that guessing never costs more than a batch per keyframe plus one per window
    assert calls <= n_key + n // window + failed, (calls, n_key, held, failed)
    assert trk1.speculation()[0] == 0
    trk.close(); trk1.close(); flow.close(); flow1.close()


@pytest.mark.gpu
def test_push_host_equals_push_dev():
    """nik_tracker_push_host -- host frames in windows, window k+1 uploaded on the context's upload stream while window k is
    registered -- gives exactly the outputs of push_dev over resident frames and of one push_u8 per frame (the reference's
    per-frame loop, main.cpp:51-86), from pageable memory, from pinned memory and with a padded row stride"""
    import ctypes as C
    import torch
    N = nik()
    geom = SMALL
    H, W = geom["H"], geom["W"]
    n, window = 70, 16                                              # (not a multiple of the window: a short last one)
    cv = synth.canvas(77, H, W)
    frames = np.stack([synth.window(cv, H, W, (i % 20) - 10, 2 * (i % 20) - 20, 0.5 * (i % 5)) for i in range(n)])
    cfg = N.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    tc = N.tracker_config(fx=600.0 * W / 640, fy=600.0 * W / 640, cx=W / 2 - 3.5, cy=H / 2 + 2.25, height=0.1,
                          max_distance=0.1, max_angle=0.02, lower_response_thr=8.0, upper_response_thr=9.0)
    keys = ("frame_id", "inserted", "good_tracking", "key_frame_id", "response", "cf_pose", "robot_pose", "distance")

    def fresh(mb):
        flow = N.CorrelationFlow(cfg, H, W, max_batch=mb, max_frames=n + window + 2)
        return flow, N.Tracker(flow, tc)
    d = torch.from_numpy(frames).cuda(); torch.cuda.synchronize()
    flow, trk = fresh(window)
    want = []
    for b in range(0, n, window):
        want += trk.push_dev(d[b:b + window].data_ptr(), min(window, n - b))
    trk.close(); flow.close()
    assert sum(o["inserted"] for o in want) >= 4
    # the next window's spectra started before the current window is pushed (nik_tracker_prefetch_dev): same outputs
    flow, trk = fresh(window)
    got_prefetch = []
    for b in range(0, n, window):
        m = min(window, n - b)
        if b + m < n:
            trk.prefetch_dev(d[b + m:b + m + window].data_ptr(), min(window, n - b - m))
        got_prefetch += trk.push_dev(d[b:b + m].data_ptr(), m)
    trk.close(); flow.close()
    flow, trk = fresh(window)
    got_pageable = trk.push_host(frames)
    trk.close(); flow.close()
    pin = torch.from_numpy(frames).pin_memory()
    flow, trk = fresh(window)
    got_pinned = trk.push_host(frames, ptr=pin.data_ptr())
    trk.close(); flow.close()
    # padded rows and frames (a cv::Mat ROI): through the C entry point directly
    pad = np.zeros((n, H + 3, W + 24), np.uint8); pad[:, :H, :W] = frames
    flow, trk = fresh(window)
    out = (N.NikTrackOutput * n)()
    rc = trk._L.nik_tracker_push_host(trk._t, n, pad.ctypes.data_as(C.c_void_p), W + 24, (H + 3) * (W + 24), C.cast(out, C.c_void_p))
    assert rc == 0
    got_padded = [o.as_dict() for o in out]
    trk.close(); flow.close()
    flow, trk = fresh(1)
    got_single = [trk.push_u8(f) for f in frames[:24]]
    trk.close(); flow.close()
    for name, got in (("prefetch", got_prefetch), ("pageable", got_pageable), ("pinned", got_pinned), ("padded", got_padded), ("push_u8", got_single)):
        for a, b in zip(got, want):
            for k in keys:
                assert a[k] == b[k], (name, a["frame_id"], k, a[k], b[k])
