"""Camera undistortion -- the step right before the KCC path (reference src/camera.cc:45-47 map construction,
:92-93 cv::remap; called from MapBuilder::AddNewInput, src/map_builder.cc:31-33).

CPU: the C oracle against the committed golden fixtures (independent numpy restatement) and live against that
restatement; the product's host map builder (nik_camera_maps, no GPU needed) against the oracle, bit-exact.
GPU: Camera::UndistortImage on device and the remap fused into the u8 -> f32 conversion, bit-exact against the
oracle; a tracked pair on raw (distorted) frames equals the oracle's pose on the undistorted frames.
"""
import json
import os
import zlib

import numpy as np
import pytest

import synth
from kcc_helpers import FULL, SMALL, check_pose_parity, nik
from oracle import kcc_oracle as O
from oracle import np_restatement as npr

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "camera_golden.json")))["cameras"]


def frame(seed, H, W):
    return synth.window(synth.canvas(seed, H, W), H, W, 0, 0)


@pytest.mark.parametrize("g", GOLD, ids=[g["name"] for g in GOLD])
def test_oracle_matches_golden(g):
    W, H = g["W"], g["H"]
    newK = O.optimal_new_camera_matrix(g["K"], g["D"], W, H)
    np.testing.assert_allclose(newK, g["new_K"], rtol=0, atol=0)
    m1, m2 = O.undistort_maps(g["K"], g["D"], newK, W, H)
    assert zlib.crc32(m1.tobytes()) == g["map1_crc"] and zlib.crc32(m2.tobytes()) == g["map2_crc"]
    und = O.remap_u8(frame(g["frame_seed"], H, W), m1, m2)
    assert zlib.crc32(und.tobytes()) == g["undistorted_crc"]
    for s in g["samples"]:
        assert (int(m1[s["r"], s["c"], 0]), int(m1[s["r"], s["c"], 1]), int(m2[s["r"], s["c"]]), int(und[s["r"], s["c"]])) == \
               (s["sx"], s["sy"], s["frac"], s["px"])


def test_oracle_matches_numpy_live():
    rng = np.random.default_rng(5)
    for _ in range(4):
        W, H = (int(v) for v in rng.choice([64, 80, 96, 120], 2))
        K = (rng.uniform(0.6, 1.1) * W, W / 2 + rng.uniform(-3, 3), rng.uniform(0.6, 1.1) * W, H / 2 + rng.uniform(-3, 3))
        D = (rng.uniform(-0.3, 0.15), rng.uniform(-0.05, 0.1), rng.uniform(-1e-3, 1e-3), rng.uniform(-1e-3, 1e-3), rng.uniform(-0.02, 0.02))
        a = O.optimal_new_camera_matrix(K, D, W, H); b = npr.optimal_new_camera_matrix(K, D, W, H)
        assert np.array_equal(a, b)
        m1, m2 = O.undistort_maps(K, D, a, W, H); n1, n2 = npr.undistort_maps(K, D, a, W, H)
        assert np.array_equal(m1, n1) and np.array_equal(m2, n2)
        img = rng.integers(0, 256, (H, W), dtype=np.uint8)
        assert np.array_equal(O.remap_u8(img, m1, m2), npr.remap_u8(img, m1, m2))


def test_known_answers():
    W, H = 80, 60
    # no distortion and new_K = K: the identity map, remap returns the image
    K = (70.0, 40.0, 70.0, 30.0); D = (0.0,) * 5
    m1, m2 = O.undistort_maps(K, D, K, W, H)
    cc, rr = np.meshgrid(np.arange(W), np.arange(H))
    assert np.array_equal(m1[..., 0], cc) and np.array_equal(m1[..., 1], rr) and not m2.any()
    img = frame(3, H, W)
    assert np.array_equal(O.remap_u8(img, m1, m2), img)
    # a pure half-pixel shift: average of the two horizontal neighbours, rounded half up; border taps are 0
    m1s = m1.copy(); m2s = np.full((H, W), 16, np.uint16)      # fx = 16/32, fy = 0
    out = O.remap_u8(img, m1s, m2s)
    nxt = np.concatenate([img[:, 1:], np.zeros((H, 1), np.uint8)], axis=1).astype(np.int64)
    assert np.array_equal(out, ((img.astype(np.int64) + nxt) * 16384 + 16384 >> 15).astype(np.uint8))
    # a map pointing fully outside gives the border value
    m1o = np.full((H, W, 2), -5, np.int16)
    assert not O.remap_u8(img, m1o, np.zeros((H, W), np.uint16)).any()
    # barrel distortion at alpha = 0: every destination pixel samples inside the source image
    g = GOLD[0]
    nk = O.optimal_new_camera_matrix(g["K"], g["D"], g["W"], g["H"])
    a1, _ = O.undistort_maps(g["K"], g["D"], nk, g["W"], g["H"])
    assert a1[..., 0].min() >= 0 and a1[..., 0].max() <= g["W"] - 1 and a1[..., 1].min() >= 0 and a1[..., 1].max() <= g["H"]


@pytest.mark.parametrize("g", GOLD, ids=[g["name"] for g in GOLD])
def test_product_host_maps_match_oracle(g):
    """nik_camera_maps is host code (no GPU): bit-identical maps and new_K."""
    N = nik()
    newK, m1, m2 = N.camera_maps(g["K"], g["D"], g["W"], g["H"])
    ok = O.optimal_new_camera_matrix(g["K"], g["D"], g["W"], g["H"])
    assert np.array_equal(newK, ok)
    o1, o2 = O.undistort_maps(g["K"], g["D"], ok, g["W"], g["H"])
    assert np.array_equal(m1, o1) and np.array_equal(m2, o2)
    assert zlib.crc32(m1.tobytes()) == g["map1_crc"] and zlib.crc32(m2.tobytes()) == g["map2_crc"]


def test_product_host_maps_reject_bad_input():
    N = nik()
    with pytest.raises(N.NikError):
        N.camera_maps((0.0, 1.0, 1.0, 1.0), (0,) * 5, 64, 48)


# ------------------------------------------------------------------------------------------------ GPU
def _cam(geom):
    for g in GOLD:
        if g["W"] == geom["W"] and g["H"] == geom["H"] and "barrel" in g["name"]:
            return g
    raise KeyError


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [SMALL, FULL], ids=["80x60", "640x480"])
def test_undistort_dev_bit_exact(geom):
    import torch
    N = nik()
    g = _cam(geom); H, W = geom["H"], geom["W"]
    _, m1, m2 = N.camera_maps(g["K"], g["D"], W, H)
    cf = N.CorrelationFlow(N.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"]), H, W, max_batch=4, max_frames=8)
    raw = np.stack([frame(10 + i, H, W) for i in range(3)])
    d_in = torch.from_numpy(raw).cuda(); d_out = torch.empty_like(d_in)
    with pytest.raises(N.NikError):
        cf.undistort_dev(d_in.data_ptr(), 3, d_out.data_ptr())          # no maps installed yet
    cf.set_undistort(m1, m2)
    cf.undistort_dev(d_in.data_ptr(), 3, d_out.data_ptr())
    got = d_out.cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], O.remap_u8(raw[i], m1, m2))
    # adversarial maps: taps on and beyond every border
    rng = np.random.default_rng(1)
    a1 = np.stack([rng.integers(-3, W + 3, (H, W)), rng.integers(-3, H + 3, (H, W))], axis=-1).astype(np.int16)
    a2 = rng.integers(0, 1024, (H, W)).astype(np.uint16)
    cf.set_undistort(a1, a2)
    cf.undistort_dev(d_in.data_ptr(), 3, d_out.data_ptr())
    assert np.array_equal(d_out.cpu().numpy()[1], O.remap_u8(raw[1], a1, a2))


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [SMALL, FULL], ids=["80x60", "640x480"])
def test_fused_undistort_intermedium_and_pose(geom):
    import torch
    N = nik()
    g = _cam(geom); H, W = geom["H"], geom["W"]
    cfg = O.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    _, m1, m2 = N.camera_maps(g["K"], g["D"], W, H)
    cf = N.CorrelationFlow(N.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"]), H, W, max_batch=4, max_frames=8)
    ora = O.Oracle(cfg, H, W)
    sh = 3 if H < 100 else 17
    key_raw, cur_raw = synth.make_pair(42, H, W, sh, -sh + 1, 2.5)
    cf.set_undistort(m1, m2)
    # single image: the stored plane is ConvertMatToNormalizedArray(UndistortImage(raw)), bit for bit
    cf.intermedium_u8(key_raw, 0)
    img, F, P = cf.frame_export(0)
    key_und = O.remap_u8(key_raw, m1, m2); cur_und = O.remap_u8(cur_raw, m1, m2)
    assert np.array_equal(img, O.Oracle.normalize_u8(key_und))
    # tracked pair on raw frames == oracle on undistorted frames
    d_cur = torch.from_numpy(cur_raw[None]).cuda()
    res = cf.track_batch_dev(d_cur.data_ptr(), [0], [1], True)
    kf, kp = ora.intermedium(O.Oracle.normalize_u8(key_und))
    ci = O.Oracle.normalize_u8(cur_und)
    cf_, cp = ora.intermedium(ci)
    pose, info, dbg = ora.compute_pose(kf, ci, kp, cp, True)
    ok, _, msg = check_pose_parity(res[0].as_dict(), pose, info, dbg, geom["PD"])
    assert ok, msg
    # removing the maps restores the plain conversion
    cf.set_undistort(None)
    cf.intermedium_u8(key_raw, 2)
    assert np.array_equal(cf.frame_export(2)[0], O.Oracle.normalize_u8(key_raw))
