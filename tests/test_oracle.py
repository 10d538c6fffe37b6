"""CPU tests: pin the C oracle (oracle/kcc_oracle.c) against
 (1) analytic known-answer tests,
 (2) the committed golden fixtures (tests/golden/kcc_golden.json, produced by the independent
     numpy/scipy restatement -- see tests/golden/make_golden.py),
 (3) that restatement run live on small geometries.
The reference ships no tests or vectors, so this is what pins the oracle ("parity unpinned").
"""
import json
import math
import os

import numpy as np
import pytest

import synth
from kcc_helpers import FULL, SMALL, ang_diff
from oracle import kcc_oracle as ko
from oracle import np_restatement as npr

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kcc_golden.json")))


def _oracle(geom, kernel=0):
    cfg = ko.default_config(kernel=kernel, rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    return ko.Oracle(cfg, geom["H"], geom["W"]), cfg


# ---------------------------------------------------------------- analytic KATs
def test_target_fft_is_checkerboard():
    """GetTargetFFT (correlation_flow.cc:46-51): FFT of a unit impulse at (rows/2, cols/2) == (-1)^(k+l)."""
    orc, _ = _oracle(SMALL)
    H, W = SMALL["H"], SMALL["W"]
    x = np.zeros((W, H), np.float32)
    x[W // 2, H // 2] = 1
    xf = orc.fft(x)
    l, k = np.meshgrid(np.arange(W), np.arange(H // 2 + 1), indexing="ij")
    assert np.abs(xf - (-1.0) ** (k + l)).max() < 1e-6


@pytest.mark.parametrize("rows,cols", [(6, 4), (60, 80), (120, 80), (448, 448), (480, 640), (720, 480)])
def test_fft_roundtrip_and_scipy(rows, cols):
    orc, _ = _oracle(SMALL)
    x = np.random.default_rng(rows * cols).random((cols, rows), dtype=np.float32)
    xf = orc.fft(x)
    ref = npr.fft(x)
    assert np.abs(xf - ref).max() / np.abs(ref).max() < 5e-7
    assert np.abs(orc.ifft(xf) - x).max() < 2e-6


def test_fft_golden_vector():
    v = GOLDEN["fft_vectors"][0]
    orc, _ = _oracle(SMALL)
    x = np.array(v["x"], np.float32).reshape(v["cols"], v["rows"])
    xf = orc.fft(x).reshape(-1)
    assert np.abs(xf.real - np.array(v["xf_re"])).max() < 1e-5
    assert np.abs(xf.imag - np.array(v["xf_im"])).max() < 1e-5


def test_ifft_ignores_imag_of_dc_and_nyquist_rows():
    """FFTW c2r semantics the restatement keeps: imaginary parts of the k=0 and k=rows/2 bins are not used
    once the column transform is done -> a spectrum of a real image round-trips exactly."""
    orc, _ = _oracle(SMALL)
    x = np.random.default_rng(5).random((SMALL["W"], SMALL["H"]), dtype=np.float32)
    assert np.abs(orc.ifft(orc.fft(x)) - x).max() < 2e-6


def test_remove_zero_component_quirk():
    """correlation_flow.cc:79-87: both assignments read the ORIGINAL x; (0,0) comes from the second one."""
    x = np.arange(5 * 4, dtype=np.float32).reshape(5, 4) ** 1.5      # cols=5, rows=4
    y = ko.Oracle.remove_zero(x)
    assert np.array_equal(y, npr.remove_zero(x))
    assert y[0, 0] == (x[1, 0] + x[4, 0]) / 2          # (x(0,1) + x(0,cols-1))/2
    assert y[2, 0] == (x[2, 1] + x[2, 3]) / 2          # row 0 of column 2: (x(1,c) + x(rows-1,c))/2
    assert np.array_equal(y[1:, 1:], x[1:, 1:])


def test_fftshift_index_map():
    x = np.random.default_rng(0).random((6, 8), dtype=np.float32)     # cols=6, rows=8
    y = ko.Oracle.fftshift(x)
    for c in range(6):
        for r in range(8):
            assert y[c, r] == x[(c - 3) % 6, (r - 4) % 8]


def test_normalize_degree():
    L = ko.lib()
    for a, want in [(0, 0), (180, -180), (-180, -180), (179.5, 179.5), (360, 0), (540, -180), (-190, 170)]:
        assert L.ora_normalize_degree(float(a)) == want


def test_rotate_zero_is_identity_and_180_is_flip():
    x = np.random.default_rng(3).random((SMALL["W"], SMALL["H"]), dtype=np.float32)
    assert np.array_equal(ko.Oracle.rotate(x, 0.0), x)
    r = ko.Oracle.rotate(x, 180.0)
    # rotation by 180 deg about (W/2, H/2) with wrap: dst(r,c) = src(H-r, W-c) (indices mod size), up to 1/32 px weights
    flip = np.roll(x[::-1, ::-1], (1, 1), axis=(0, 1))
    assert np.abs(r - flip).max() < 1e-4


def test_psr_of_impulse_response():
    """GetInfo (correlation_flow.cc:238-243) on a pure impulse: mean=0, std=sqrt(1/n)."""
    n = 4096
    g = np.zeros(n, np.float32)
    g[17] = 1
    assert abs(ko.Oracle.get_info(g, 1.0) - 1.0 / (math.sqrt(1.0 / n) + 1e-7)) < 1e-2


def test_invalid_kernel_id_raises():
    cfg = ko.default_config(kernel=5, rotation_divisor=SMALL["PD"], rotation_channel=SMALL["PC"])
    orc = ko.Oracle(cfg, SMALL["H"], SMALL["W"])
    z = np.zeros((SMALL["W"], SMALL["H"] // 2 + 1), np.complex64)
    with pytest.raises(ValueError, match="invalid kernel"):
        orc.estimate_trans(z, z, 0)


def test_odd_height_rejected():
    with pytest.raises(ValueError):
        ko.Oracle(ko.default_config(), 61, 80)


@pytest.mark.parametrize("dy,dx", [(0, 0), (5, -7), (-6, 4)])
def test_cyclic_shift_known_answer(dy, dx):
    """A cyclic shift of the image content by (-dy,-dx) gives trans=(dy,dx) exactly and pose=(dx,dy,0)."""
    geom = SMALL
    orc, _ = _oracle(geom)
    key, _ = synth.make_pair(11, geom["H"], geom["W"], 0, 0)
    ki = orc.normalize_u8(key)
    ci = np.roll(ki, (-dx, -dy), axis=(0, 1))
    kf, kp = orc.intermedium(ki)
    cf, cp = orc.intermedium(ci)
    pose, info, dbg = orc.compute_pose(kf, ci, kp, cp, True)
    assert (pose[0], pose[1], pose[2]) == (dx, dy, 0.0)
    assert info[0] > 20


# ---------------------------------------------------------------- golden fixtures
def test_gather_golden():
    case = GOLDEN["gather_cases"][0]
    g = GOLDEN["geoms"][case["geom"]]
    orc, _ = _oracle(g)
    plane = np.random.default_rng(case["seed"]).random((g["W"], g["H"]), dtype=np.float32)
    pol = orc.polar(orc.fftshift(orc.remove_zero(plane)))
    assert abs(float(pol.sum(dtype=np.float64)) - case["polar_sum"]) < 1e-9 * abs(case["polar_sum"])
    for (j, i), v in zip([(0, 0), (5, 7), (40, 60), (79, 119), (79, 30)], case["polar_samples"]):
        assert float(pol[j, i]) == v
    for rc in case["rotations"]:
        r = ko.Oracle.rotate(plane, rc["degree"])
        assert abs(float(r.sum(dtype=np.float64)) - rc["sum"]) < 1e-9 * abs(rc["sum"])
        for (c, rr), v in zip([(0, 0), (3, 9), (40, 30), (79, 59)], rc["samples"]):
            assert float(r[c, rr]) == v


@pytest.mark.parametrize("case", GOLDEN["pose_cases"], ids=lambda c: "%s-%d" % (c["geom"], c["seed"]))
def test_pose_golden(case):
    g = GOLDEN["geoms"][case["geom"]]
    orc, _ = _oracle(g, case["kernel"])
    key, cur = synth.make_pair(case["seed"], g["H"], g["W"], case["dy"], case["dx"], case["theta"])
    ki, ci = orc.normalize_u8(key), orc.normalize_u8(cur)
    kf, kp = orc.intermedium(ki)
    cf, cp = orc.intermedium(ci)
    cs = [float(np.abs(cf).sum(dtype=np.float64)), float(np.abs(cp).sum(dtype=np.float64))]
    assert abs(cs[0] - case["spectrum_checksum"][0]) < 1e-5 * cs[0]
    assert abs(cs[1] - case["spectrum_checksum"][1]) < 1e-4 * cs[1]
    pose, info, dbg = orc.compute_pose(kf, ci, kp, cp, case["small_rot"])
    PD = g["PD"]
    if case["rot_gap"] > 1e-3:       # clear winner between the two mirror peaks -> indices must be identical
        assert (dbg["rot_row"], dbg["rot_col"]) == (case["rot_row"], case["rot_col"])
        assert dbg["chosen"] == case["chosen"]
        assert pose[2] == pytest.approx(case["pose"][2], abs=1e-7)
    else:                            # rounding-noise tie of the 180-degree mirror: same rotation mod 180
        assert dbg["rot_col"] == case["rot_col"] and (dbg["rot_row"] - case["rot_row"]) % (PD // 2) == 0
    assert (pose[0], pose[1]) == (case["pose"][0], case["pose"][1])
    assert ang_diff(pose[2], case["pose"][2]) < 1e-6
    ch = dbg["chosen"]
    assert (dbg["trans_row"][ch], dbg["trans_col"][ch]) == (-int(pose[1]) + g["H"] // 2, -int(pose[0]) + g["W"] // 2)
    assert info[0] == pytest.approx(case["info"][0], rel=1e-3) and info[2] == pytest.approx(case["info"][2], rel=1e-3)


# ---------------------------------------------------------------- live cross-check
def test_oracle_vs_numpy_live_small():
    g = SMALL
    for kernel in (0, 1):
        orc, _ = _oracle(g, kernel)
        nr = npr.CorrelationFlowNp(g["H"], g["W"], g["PD"], g["PC"], kernel=kernel)
        key, cur = synth.make_pair(21, g["H"], g["W"], 4, -3, 6.0)
        ki, ci = orc.normalize_u8(key), orc.normalize_u8(cur)
        assert np.array_equal(ki, key.T.astype(np.float32) / np.float32(255))
        kf, kp = orc.intermedium(ki)
        kf2, kp2 = nr.intermedium(ki)
        assert np.abs(kf - kf2).max() / np.abs(kf2).max() < 1e-6
        assert np.abs(kp - kp2).max() / np.abs(kp2).max() < 1e-5
        assert np.array_equal(orc.polar(ki), npr.polar(ki, g["PD"], g["PC"]))
        for deg in (-7.5, 33.0, 181.5):
            assert np.array_equal(ko.Oracle.rotate(ci, deg), npr.rotate(ci, deg))
        cf, cp = orc.intermedium(ci)
        for sr in (True, False):
            p1, i1, d1 = orc.compute_pose(kf, ci, kp, cp, sr)
            p2, i2, d2 = nr.compute_pose(kf2, ci, *nr.intermedium(ci)[::-1][:1], cp, sr) if False else nr.compute_pose(kf2, ci, kp2, nr.intermedium(ci)[1], sr)
            assert list(p1[:2]) == list(p2[:2]) and ang_diff(p1[2], p2[2]) < 1e-6
            assert i1 == pytest.approx(i2, rel=2e-3)


def test_track_pairs_threads_agree():
    g = SMALL
    _, cfg = _oracle(g)
    keys, curs, _ = synth.make_batch(4, g["H"], g["W"], seed0=900)
    a = ko.track_pairs(cfg, keys, curs, True, nthreads=1)
    b = ko.track_pairs(cfg, keys, curs, True, faithful=True, nthreads=2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ---- OpenCV semantics by construction (tests/opencv_kats.py): right-angle rotations are permutations, the polar transform's
# ---- four axes are two-tap blends at exact 1/32-pixel positions.  No OpenCV build, no recalled rounding rule involved.
@pytest.mark.parametrize("geom", [SMALL, FULL])
def test_rotate_right_angles_are_permutations(geom):
    import opencv_kats as kat
    x = np.random.default_rng(11).random((geom["W"], geom["H"]), dtype=np.float32)
    for deg, q in [(0.0, 0), (90.0, 1), (180.0, 2), (270.0, 3), (-90.0, 3), (-180.0, 2), (360.0, 0), (-270.0, 1), (450.0, 1)]:
        assert np.array_equal(ko.Oracle.rotate(x, deg), kat.rotate_right_angle(x, q)), "RotateArray(%g deg)" % deg


@pytest.mark.parametrize("geom", [SMALL, FULL])
def test_polar_axes_known_answers(geom):
    import opencv_kats as kat
    orc = ko.Oracle(ko.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"]), geom["H"], geom["W"])
    x = np.random.default_rng(12).integers(0, 256, (geom["W"], geom["H"])).astype(np.float32)
    got = orc.polar(x)                                          # (PC, PD): got[j, i] = radius j, angle row i
    for i, want in kat.polar_axes(x, geom["PD"], geom["PC"]).items():
        assert np.array_equal(got[:, i], want), "warpPolar angle row %d" % i
    # RemoveZeroComponent + fftshift, restated in numpy, feed the same transform (what ComputeIntermedium does, :93-94)
    assert np.array_equal(orc.fftshift(orc.remove_zero(x)), kat.remove_zero_fftshift(x))
