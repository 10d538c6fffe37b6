"""Vectors produced by the REAL reference (oracle/pin/pin_reference: sair-lab/ni-slam's own correlation_flow.cc + utils.cc
with FFTW3f, Eigen 3 and OpenCV 4.x; see oracle/pin/README.md) against the CPU oracle and -- with -m gpu -- the HIP path.

The image this repository is developed in has none of those libraries, so the files tests/golden/ref_*.json / ref_*.bin do
not exist yet and every test here SKIPS ("parity unpinned").  Whoever has the reference's dependencies creates
them with the three commands of oracle/pin/README.md; from then on these tests pin the oracle for good.
Inputs never travel: the pairs are regenerated from the seeds of oracle/pin/make_inputs.py, the RECALLED.md experiments
from a 31-bit LCG that both sides implement."""
import json
import os
import sys

import numpy as np
import pytest

from kcc_helpers import ROOT, ang_diff, nik
from oracle import kcc_oracle as O

GOLD = os.environ.get("NIK_REF_GOLDEN_DIR") or os.path.join(ROOT, "tests", "golden")     # (override: oracle/pin/dry_run.py)
sys.path.insert(0, os.path.join(ROOT, "oracle", "pin"))


def _need(*names):
    paths = [os.path.join(GOLD, n) for n in names]
    if not all(os.path.exists(p) for p in paths):
        pytest.skip("reference vectors absent (oracle/pin/README.md): parity unpinned")
    return paths


def lcg_state(n, seed):
    s = np.empty(n, np.uint32); v = np.uint32(seed)
    with np.errstate(over="ignore"):
        for i in range(n):
            v = v * np.uint32(1103515245) + np.uint32(12345)
            s[i] = v
    return s


def lcg_floats(rows, cols, seed):
    """pin_reference.cpp lcg_array: plane(r, c) filled row by row with 24-bit fractions (exact in float32).  Returned in the
    oracle's array convention: shape (cols, rows), element [c, r] -- the memory image of a column-major rows x cols Eigen array"""
    s = lcg_state(rows * cols, seed)
    return np.ascontiguousarray((((s >> 8) & 0xFFFFFF).astype(np.float32) / np.float32(16777216.0)).reshape(rows, cols).T)


def lcg_bytes(n, seed):
    return ((lcg_state(n, seed) >> 16) & 0xFF).astype(np.uint8)


def test_lcg_is_the_documented_one():
    assert lcg_state(3, 12345).tolist() == [(12345 * 1103515245 + 12345) % 2**32,
                                            (((12345 * 1103515245 + 12345) % 2**32) * 1103515245 + 12345) % 2**32,
                                            ((((12345 * 1103515245 + 12345) % 2**32) * 1103515245 + 12345) % 2**32 * 1103515245 + 12345) % 2**32]
    x = lcg_floats(2, 3, 12345)
    assert x.dtype == np.float32 and x.shape == (3, 2) and 0 <= x.min() and x.max() < 1


def _colmajor(path, rows, cols, dtype=np.float32):
    return np.fromfile(path, dtype).reshape(cols, rows)              # Eigen storage order == the oracle's (cols, rows) arrays


# ---- RECALLED.md experiments --------------------------------------------------------------------------------------

def test_recalled_gathers_bit_exact():
    """#1-#6: cv::warpPolar and RotateArray's warpAffine on the real OpenCV == the oracle, bit for bit"""
    (meta,) = _need("ref_recalled.json")
    m = json.load(open(meta))
    H, W, PD, PC = m["H"], m["W"], m["PD"], m["PC"]
    x = lcg_floats(H, W, 12345)
    ora = O.Oracle(O.default_config(rotation_divisor=PD, rotation_channel=PC), H, W)
    (pp,) = _need("ref_polar.bin")
    assert np.array_equal(ora.polar(x), _colmajor(pp, PD, PC))
    for k, deg in enumerate(m["rotate_degrees"]):
        (rp,) = _need("ref_rotate_%d.bin" % k)
        assert np.array_equal(O.Oracle.rotate(x, float(deg)), _colmajor(rp, H, W)), deg


def test_recalled_fft_layout_and_values():
    """#12-#14: FFTW3f r2c / c2r through the reference's own FFT / IFFT: layout exact, values to float32 rounding"""
    meta, ff, fi, fn = _need("ref_recalled.json", "ref_fft.bin", "ref_ifft.bin", "ref_ifft_nonhermitian.bin")
    m = json.load(open(meta)); H, W = m["H"], m["W"]
    x = lcg_floats(H, W, 12345)
    ora = O.Oracle(O.default_config(rotation_divisor=m["PD"], rotation_channel=m["PC"]), H, W)
    ref = np.fromfile(ff, np.complex64).reshape(W, H // 2 + 1)
    got = ora.fft(x)
    assert np.abs(got - ref).max() <= 3e-6 * np.abs(ref).max()
    assert np.abs(ora.ifft(ref) - _colmajor(fi, H, W)).max() <= 3e-6
    bad = ref.copy(); bad[:, 0] += 3.5j; bad[:, H // 2] += -2.25j
    assert np.abs(ora.ifft(bad) - _colmajor(fn, H, W)).max() <= 3e-6      # #13: imaginary DC / Nyquist rows are ignored


def test_recalled_camera_and_colour():
    """#8-#11: getOptimalNewCameraMatrix, initUndistortRectifyMap(CV_16SC2), remap on u8, cvtColor(RGB2GRAY)"""
    meta, p1, p2, pr, pg = _need("ref_recalled.json", "ref_map1.bin", "ref_map2.bin", "ref_remap_u8.bin", "ref_rgb2gray.bin")
    m = json.load(open(meta)); H, W = m["H"], m["W"]
    K, D = m["camera"]["K"], m["camera"]["D"]
    newK = O.optimal_new_camera_matrix(K, D, W, H)
    assert np.allclose(newK, m["camera"]["new_K"], rtol=1e-12, atol=0)
    m1, m2 = O.undistort_maps(K, D, newK, W, H)
    assert np.array_equal(m1.reshape(-1), np.fromfile(p1, np.int16)) and np.array_equal(m2.reshape(-1), np.fromfile(p2, np.uint16))
    img = lcg_bytes(H * W, 999).reshape(H, W)
    assert np.array_equal(O.remap_u8(img, m1, m2).reshape(-1), np.fromfile(pr, np.uint8))
    rgb = lcg_bytes(3 * H * W, 4242).reshape(H, W, 3).astype(np.int64)
    gray = ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(gray.reshape(-1), np.fromfile(pg, np.uint8))


def test_recalled_eigen_rules():
    """#15 maxCoeff tie-break (first maximum in column-major order), #16 Array::pow(int) == double cube rounded to float"""
    meta, pp = _need("ref_recalled.json", "ref_pow3.bin")
    m = json.load(open(meta)); H, W = m["H"], m["W"]
    t = m["maxcoeff_tie"]
    first = min(t["set"], key=lambda rc: rc[1] * H + rc[0])
    assert [t["row"], t["col"]] == first
    x = lcg_floats(H, W, 12345)
    b = (x + np.float32(0.1)).astype(np.float64)
    assert np.array_equal((b * b * b).astype(np.float32), _colmajor(pp, H, W))


def test_recalled_estimate_trans():
    (meta,) = _need("ref_recalled.json")
    m = json.load(open(meta)); H, W = m["H"], m["W"]
    ora = O.Oracle(O.default_config(rotation_divisor=m["PD"], rotation_channel=m["PC"]), H, W)
    x, z = lcg_floats(H, W, 12345), lcg_floats(H, W, 777)
    psr, trans = ora.estimate_trans(ora.fft(z), ora.fft(x), 0)[:2]
    assert list(trans) == m["estimate_trans"]["trans"]
    assert abs(psr - m["estimate_trans"]["psr"]) <= 2e-3 * abs(m["estimate_trans"]["psr"])


# ---- whole-path vectors ----------------------------------------------------------------------------------------------

def _cases():
    (p,) = _need("ref_pairs.json")
    import make_inputs
    ref = {c["name"]: c for c in json.load(open(p))["cases"]}
    for case in make_inputs.CASES:
        if case[0] in ref:
            yield case, ref[case[0]], make_inputs.pairs_of(case)


def _check(got_pose, got_info, want, what):
    assert got_pose[0] == want["pose"][0] and got_pose[1] == want["pose"][1], (what, got_pose, want["pose"])
    assert ang_diff(got_pose[2], want["pose"][2]) <= 1e-4, (what, got_pose, want["pose"])            # north_star: 1e-4 rad
    for k in (0, 2):
        assert abs(got_info[k] - want["info"][k]) <= 5e-3 * abs(want["info"][k]), (what, got_info, want["info"])


def test_oracle_matches_the_reference_on_pairs():
    for (name, H, W, PD, PC, n, seed0, mt), ref, (keys, curs, _) in _cases():
        ocfg = O.default_config(rotation_divisor=PD, rotation_channel=PC)
        poses, infos, _, _ = O.track_pairs(ocfg, keys, curs, bool(ref["not_large_rotation"]), nthreads=4)
        for i in range(n):
            _check(list(poses[i]), list(infos[i]), ref["pairs"][i], (name, i))
        ora = O.Oracle(ocfg, H, W)
        kf, kp = ora.intermedium(ora.normalize_u8(keys[0]))
        pr = ref["pairs"][0]
        assert abs(np.abs(kf).sum() - pr["F_abs_sum"]) <= 1e-5 * pr["F_abs_sum"] and abs(np.abs(kp).sum() - pr["P_abs_sum"]) <= 1e-4 * pr["P_abs_sum"]
        assert np.allclose([kf[0, 0].real, kf[2, 1].real, kf[2, 1].imag, kf[W - 1, H // 2].real], pr["F_probe"], rtol=1e-5, atol=1e-5 * abs(kf[0, 0]))


@pytest.mark.gpu
def test_hip_path_matches_the_reference_on_pairs():
    import torch
    N = nik()
    for (name, H, W, PD, PC, n, seed0, mt), ref, (keys, curs, _) in _cases():
        cf = N.CorrelationFlow(N.default_config(rotation_divisor=PD, rotation_channel=PC), H, W, max_batch=n, max_frames=2 * n)
        dk, dc = torch.from_numpy(keys).cuda(), torch.from_numpy(curs).cuda(); torch.cuda.synchronize()
        cf.intermedium_batch_dev(dk.data_ptr(), n, list(range(n)))
        res = cf.track_batch_dev(dc.data_ptr(), list(range(n)), list(range(n, 2 * n)), bool(ref["not_large_rotation"]))
        for i in range(n):
            r = res[i].as_dict()
            _check(r["pose"], r["info"], ref["pairs"][i], (name, i))
        cf.close()


def test_consumer_dry_run():
    """oracle/pin/dry_run.py: the tests above run against stand-in vectors the oracle writes into a temporary directory --
    proves formats, array conventions and comparisons are consistent before anyone has built the kit (it pins nothing)."""
    if os.environ.get("NIK_REF_GOLDEN_DIR"):
        pytest.skip("already inside the dry run")
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "pin", "dry_run.py"), "-m", "not gpu", "-k", "not dry_run"],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and "passed" in r.stdout and "skipped" not in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
