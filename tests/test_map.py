"""Keyframe map and loop-closure candidate management (reference src/map.cc, src/loop_closure.cc): the C++ host code
ni-slam_amd/csrc/kcc_map.cpp against the Python restatement tests/ref_map.py.  Candidate selection is host-only logic
(CPU tests); the GPU test checks a whole FindLoopClosure -- candidates, batched registrations, winner and `found` --
against the oracle."""
import ctypes

import numpy as np
import pytest

import synth
from kcc_helpers import SMALL, ang_diff, nik
from oracle import kcc_oracle as O
from ref_map import RefMap


def _both(N, **kw):
    ref = RefMap(**kw)
    m = N.KeyframeMap(None, N.loop_config(**kw))
    return ref, m


def test_struct_sizes():
    N = nik()
    assert ctypes.sizeof(N.NikLoopConfig) == 8 + 4 + 4 + 8 + 8 + 8
    assert ctypes.sizeof(N.NikLoopResult) == 5 * 4 + 4 + 3 * 8 + 3 * 8
    assert ctypes.sizeof(N.NikTrackOutput) == 5 * 4 + 4 + 10 * 8 + 2 * 4


def test_candidates_random_walk():
    N = nik()
    rng = np.random.default_rng(11)
    for gap, dthr in [(100, 5.0), (10, 0.5), (0, 0.0), (5, 0.0), (0, 1.0)]:
        ref, m = _both(N, grid_scale=0.1, frame_gap_thr=gap, distance_thr=dthr)
        pos = np.zeros(2); dist = 0.0; fid = 0
        for step in range(300):
            delta = rng.normal(0, 0.06, 2)
            pos = pos + delta; dist += float(np.hypot(*delta))
            fid += int(rng.integers(1, 4))                      # keyframes are a subset of the frames: ids have gaps
            pose = (pos[0], pos[1], float(rng.uniform(-3, 3)))
            d = None if (dthr == 0.0 and step % 7 == 0) else dist
            rid = ref.add_frame(fid, pose, d)
            m.add_frame(fid, step, pose, d)
            for prior in (None, pose, (pose[0] + 0.13, pose[1] - 0.08, 0.0)):
                assert m.candidates(rid, prior) == ref.candidates(rid, prior), (gap, dthr, step, prior)
        assert len(m) == len(ref.frames)


def test_grid_truncation_and_first_id():
    N = nik()
    ref, m = _both(N, grid_scale=0.1, frame_gap_thr=0, distance_thr=0.0)
    # the first frame's id is forced to 0; cells truncate toward zero, so (-0.05, 0.05) and (0.05, -0.05) share cell (0,0)
    for k, (fid, pose) in enumerate([(7, (-0.05, 0.05, 0)), (9, (0.05, -0.05, 0)), (12, (-0.15, 0.0, 0)), (15, (0.35, 0.0, 0))]):
        ref.add_frame(fid, pose); m.add_frame(fid, k, pose)
    assert sorted(ref.frames) == [0, 9, 12, 15]
    assert m.candidates(0) == [0, 9, 12, 15] == ref.candidates(0)
    assert m.candidates(9, (0.0, 0.0, 0.0)) == ref.candidates(9, (0.0, 0.0, 0.0)) == [0, 9, 12]
    assert m.candidates(9, (0.31, 0.0, 0.0)) == ref.candidates(9, (0.31, 0.0, 0.0)) == [15]
    with pytest.raises(N.NikError):
        m.add_frame(9, 5, (0, 0, 0))                            # duplicate id
    with pytest.raises(N.NikError):
        m.candidates(99)
    with pytest.raises(N.NikError):
        m.find_loop(9)                                          # no context: candidate queries only


@pytest.mark.gpu
def test_find_loop_matches_oracle():
    import torch
    N = nik()
    geom = SMALL; H, W = geom["H"], geom["W"]
    cfg = N.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    ocfg = O.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    cf = N.CorrelationFlow(cfg, H, W, max_batch=4, max_frames=32)      # max_batch < candidates: nik_match chunks
    ora = O.Oracle(ocfg, H, W)
    kw = dict(grid_scale=0.1, frame_gap_thr=3, distance_thr=0.2, position_response_thr=20.0, angle_response_thr=20.0)
    ref = RefMap(**kw); m = N.KeyframeMap(cf, N.loop_config(**kw))
    # a loop: frames 0..11 walk away over the texture, frame 12 comes back near frame 1 (rotated by 180 degrees: the
    # two-hypothesis path), frame 13 is somewhere never seen
    cv = synth.canvas(5, H, W)
    offs = [(0, 0), (2, 1), (6, 5), (11, 9), (15, 14), (19, 18), (22, 22), (24, 25), (25, 27), (26, 28), (27, 29), (28, 30)]
    frames = [synth.window(cv, H, W, dy, dx) for dy, dx in offs]
    frames.append(synth.window(cv, H, W, 3, 0, 180.0))
    frames.append(synth.window(synth.canvas(99, H, W), H, W, 0, 0))
    spectra = {}
    dist = 0.0
    for i, img in enumerate(frames):
        cf.intermedium_u8(img, i)
        f32 = ora.normalize_u8(img); spectra[i] = (f32,) + tuple(ora.intermedium(f32))
        # robot poses in metres on a 0.1 m grid: consecutive frames fall into neighbouring cells
        p = (0.02 * (offs[i][1] if i < len(offs) else (0 if i == 12 else 40)), 0.02 * (offs[i][0] if i < len(offs) else (3 if i == 12 else 40)), 0.0)
        dist += 0.1
        ref.add_frame(i, p, dist); m.add_frame(i, i, p, dist)
        for prior in (None, p):
            def cp(fid, i=i):
                pose, info, _ = ora.compute_pose(spectra[fid][1], spectra[i][0], spectra[fid][2], spectra[i][2], False)
                return pose, info
            want = ref.find_loop(i, cp, prior)
            got = m.find_loop(i, prior)
            assert got["n_candidates"] == want["n_candidates"], (i, prior)
            assert got["loop_frame_id"] == want["loop_frame_id"] and got["found"] == want["found"], (i, prior, got, want)
            if want["loop_frame_id"] >= 0:
                assert got["relative_pose"][:2] == want["relative_pose"][:2]
                assert ang_diff(got["relative_pose"][2], want["relative_pose"][2]) < 1e-6
                np.testing.assert_allclose(got["response"], want["response"], rtol=5e-3)
            else:
                assert got["response"] == [-1.0, -1.0, -1.0]
    # the revisit must have been found against an early frame, the unseen place must not
    assert m.find_loop(12)["found"] and m.find_loop(12)["loop_frame_id"] in (0, 1, 2)
    assert not m.find_loop(13)["found"]
