"""The any-size kernel family (ni-slam_amd/csrc/kcc_generic.hip): geometries the reference accepts (CorrelationFlow::FFT takes any
size with even rows, src/correlation_flow.cc:53-77; rotation_divisor / rotation_channel are free YAML values,
configs/config_geekplus.yaml:9-10) but the tiled kernels are not instantiated for -- a 752 x 480 camera (752 = 16 x 47: a prime
factor outside {2,3,5,7}), 512 x 512, polar planes 720 x 64 and 360 x 240, an aspect ratio beyond 2:1, widths that are not
multiples of 16.  Same parity bar as the tiled path: gathers bit-exact, FFTs within float32 rounding of the oracle's, arg-max
indices and poses exact (up to the documented rotation tie), PSR within 2e-3.  And at 640 x 480, where both families exist,
they must agree with each other."""
import os

import numpy as np
import pytest

import synth
from kcc_helpers import FULL, check_pose_parity, nik
from oracle import kcc_oracle as ko

pytestmark = pytest.mark.gpu

# fam: which plane families run the any-size kernels (nik_is_generic: bit 0 image, bit 1 polar) -- the choice is per family, so a
# 640 x 480 camera with a 720 x 64 polar plane keeps the tiled image kernels
# (round 5: 752 x 480 and 512 x 512 have tiled plans now -- 752 = 16 x 47 as a radix-16 pass plus a direct 47-point pass shared
# out over the line's threads, kcc_fft2.h PlanPrime; 256 = 16 x 16, 512 = 8 x 8 x 8 / 16 x 32 -- so each of them runs twice here: on
# the tiled family it now gets by default (fam 0), and forced onto the any-size family ($NIK_GENERIC=4) as before)
GEOMS = [
    pytest.param(dict(H=480, W=752, PD=720, PC=480, fam=(0,)), id="752x480-tiled"),    # EuRoC-style camera; 752 = 2^4 x 47
    pytest.param(dict(H=512, W=512, PD=720, PC=480, fam=(0, 2)), id="512x512-tiled"),
    pytest.param(dict(H=480, W=848, PD=720, PC=480, fam=(0,)), id="848x480-tiled"),    # 848 = 2^4 x 53: the same prime-factor plan
    pytest.param(dict(H=768, W=1024, PD=720, PC=480, fam=(0, 2)), id="1024x768-tiled"),
    pytest.param(dict(H=480, W=752, PD=720, PC=480, fam=(1,), force="4"), id="752x480"),
    pytest.param(dict(H=512, W=512, PD=720, PC=480, fam=(1,), force="4"), id="512x512"),
    pytest.param(dict(H=480, W=640, PD=720, PC=64, fam=(2,)), id="polar720x64"),        # config_geekplus.yaml's "64 may work well"
    pytest.param(dict(H=480, W=640, PD=360, PC=240, fam=(2,)), id="polar360x240"),
    pytest.param(dict(H=448, W=448, PD=720, PC=64, fam=(2,)), id="448x448-polar720x64"),  # config_geekplus.yaml with that suggestion
    pytest.param(dict(H=100, W=300, PD=120, PC=80, fam=(1, 3)), id="300x100-aspect3"),  # beyond 2:1: multi-period BORDER_WRAP
    pytest.param(dict(H=62, W=94, PD=90, PC=50, fam=(3,)), id="94x62"),                 # nothing a multiple of 16; 94 = 2 x 47, 62 = 2 x 31
]


def _relmax(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _mk(geom, n):
    N = nik()
    cfg = N.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    ocfg = ko.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    cf = N.CorrelationFlow(cfg, geom["H"], geom["W"], max_batch=n, max_frames=2 * n + 2)
    return N, cf, ko.Oracle(ocfg, geom["H"], geom["W"]), ocfg


@pytest.mark.parametrize("geom", GEOMS)
def test_any_size_geometry_matches_oracle(geom):
    import torch
    H, W, PD, PC = geom["H"], geom["W"], geom["PD"], geom["PC"]
    n = 6
    old = os.environ.get("NIK_GENERIC")
    if geom.get("force") and old is None:
        os.environ["NIK_GENERIC"] = geom["force"]
    try:
        N, cf, orc, ocfg = _mk(geom, n)
    finally:
        if geom.get("force") and old is None:
            os.environ.pop("NIK_GENERIC", None)
    if old is None:                                              # (under a forced outer run -- the callers' test below -- the mask is the forced one)
        assert cf._L.nik_is_generic(cf._ctx) in geom["fam"]
    rng = np.random.default_rng(H + W)
    # FFT / IFFT of both plane families against the oracle's
    for which, (rows, cols) in enumerate([(H, W), (PD, PC)]):
        x = rng.random((cols, rows), dtype=np.float32)
        ref = orc.fft(x)
        assert _relmax(cf.dbg_fft(x, which), ref) < 3e-6, "forward FFT %d x %d" % (rows, cols)
        assert np.abs(cf.dbg_ifft(ref, which) - x).max() < 3e-6, "inverse FFT %d x %d" % (rows, cols)
    # gathers: bit-exact
    x = rng.random((W, H), dtype=np.float32)
    assert np.array_equal(cf.dbg_polar(x), orc.polar(orc.fftshift(orc.remove_zero(x))))
    img = synth.canvas(3, H, W)[:H, :W]
    cf.intermedium_u8(img, 2 * n)
    xn = orc.normalize_u8(img)
    got_img, f, p = cf.frame_export(2 * n)
    assert np.array_equal(got_img, xn)
    for deg2 in (0, 1, -37, 180, 359, -719):
        assert np.array_equal(cf.dbg_rotate(2 * n, deg2), orc.rotate(xn, deg2 * 0.5)), "RotateArray(%g deg)" % (deg2 * 0.5)
    rf, rp = orc.intermedium(xn)
    assert _relmax(f, rf) < 3e-6 and _relmax(p, rp) < 3e-5
    # poses: host entry points and the batched device entry point, both ComputePose modes
    keys, curs, motions = synth.make_batch(n, H, W, seed0=4100 + H + W, max_shift=max(2, min(H, W) // 12), max_theta=8.0)
    for small in (True, False):
        poses, infos, dbgs, _ = ko.track_pairs(ocfg, keys, curs, small, nthreads=n)
        dk, dc = torch.from_numpy(keys).cuda(), torch.from_numpy(curs).cuda()
        torch.cuda.synchronize()
        cf.intermedium_batch_dev(dk.data_ptr(), n, list(range(n)))
        res = cf.track_batch_dev(dc.data_ptr(), list(range(n)), list(range(n, 2 * n)), small, sync=True)
        for i in range(n):
            def rerun(row, col, i=i):
                o = ko.Oracle(ocfg, H, W); o.force_rotation(row, col)
                kf, kp = o.intermedium(o.normalize_u8(keys[i])); ci = o.normalize_u8(curs[i]); _, cp = o.intermedium(ci)
                return o.compute_pose(kf, ci, kp, cp, small)
            ok, _, msg = check_pose_parity(res[i].as_dict(), poses[i], infos[i], dbgs[i], PD, rerun=rerun)
            assert ok, "%s pair %d %s: %s" % ("small" if small else "large", i, motions[i], msg)
    # the reference's own entry-point shapes (f32 column-major in, spectra by value) on the any-size family
    f2, p2 = cf.ComputeIntermedium(orc.normalize_u8(curs[0]), dst=2 * n + 1)
    rf, rp = orc.intermedium(orc.normalize_u8(curs[0]))
    assert _relmax(f2, rf) < 3e-6 and _relmax(p2, rp) < 3e-5
    cf.close()


def test_any_size_family_agrees_with_the_tiled_kernels_at_640x480():
    """where both families exist they must tell the same story -- also mixed (polar family on the any-size kernels, image family
    tiled, and the other way round): identical arg-max indices and poses on 32 pairs (up to the rotation tie every one of them is
    allowed against the oracle), PSR within 2e-3, spectra within float32 rounding"""
    import torch
    N = nik()
    H, W, PD, PC = FULL["H"], FULL["W"], FULL["PD"], FULL["PC"]
    n = 32
    keys, curs, motions = synth.make_batch(n, H, W, seed0=8800, max_shift=40, max_theta=10.0)
    dk, dc = torch.from_numpy(keys).cuda(), torch.from_numpy(curs).cuda()
    torch.cuda.synchronize()
    out = {}
    forced = {"tiled": None, "generic": "1", "polar-generic": "2", "image-generic": "4"}      # $NIK_GENERIC: 1 both, bit 1 polar, bit 2 image
    masks = {"tiled": 0, "generic": 3, "polar-generic": 2, "image-generic": 1}
    for fam, env in forced.items():
        if env:
            os.environ["NIK_GENERIC"] = env
        try:
            cf = N.CorrelationFlow(N.default_config(), H, W, max_batch=n, max_frames=2 * n)
        finally:
            os.environ.pop("NIK_GENERIC", None)
        assert cf._L.nik_is_generic(cf._ctx) == masks[fam]
        cf.intermedium_batch_dev(dk.data_ptr(), n, list(range(n)))
        res = [r.as_dict() for r in cf.track_batch_dev(dc.data_ptr(), list(range(n)), list(range(n, 2 * n)), True, sync=True)]
        _, f, p = cf.frame_export(n + 3)
        out[fam] = (res, f, p)
        cf.close()
    rt, ft, pt = out["tiled"]
    for fam in ("generic", "polar-generic", "image-generic"):
        _, fg, pg = out[fam]
        assert _relmax(fg, ft) < 3e-6 and _relmax(pg, pt) < 3e-5, fam
    ocfg = ko.default_config()
    poses, infos, dbgs, _ = ko.track_pairs(ocfg, keys, curs, True, nthreads=min(32, os.cpu_count() or 1))
    for i in range(n):
        for fam in forced:
            r = out[fam][0][i]
            ok, _, msg = check_pose_parity(r, poses[i], infos[i], dbgs[i], PD)
            assert ok, "%s pair %d %s: %s" % (fam, i, motions[i], msg)
            assert r["trans_row"] == rt[i]["trans_row"] and r["trans_col"] == rt[i]["trans_col"], fam
            assert r["rot_col"] == rt[i]["rot_col"] and (r["rot_row"] - rt[i]["rot_row"]) % (PD // 2) == 0, fam


@pytest.mark.timeout(1800)
def test_callers_of_the_boundary_on_the_any_size_family():
    """the sequence driver, the key-frame map / loop closure, the coarse-to-fine pyramid and the multi-GPU group run unchanged on
    top of the any-size family: their own GPU tests, re-run in a process where every context is forced onto it ($NIK_GENERIC=1)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NIK_GENERIC="1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "tests/test_tracker.py", "tests/test_map.py", "tests/test_pyramid.py",
                        "tests/test_pipeline.py", "tests/test_camera.py", "tests/test_group.py::test_local_group_of_one_gpu_equals_direct_calls"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1700)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout and "failed" not in p.stdout


def test_random_even_geometries():
    """a dozen random geometries -- even sizes from 12 to 180, whatever prime factors they have -- through ComputeIntermedium and both
    ComputePose modes: the any-size kernels (and whichever tiled family happens to fit) against the oracle"""
    N = nik()
    rng = np.random.default_rng(20240)
    done = 0
    for trial in range(14):
        H, W = (int(2 * rng.integers(6, 91)) for _ in range(2))
        PD, PC = (int(2 * rng.integers(6, 81)) for _ in range(2))
        cfg = N.default_config(rotation_divisor=PD, rotation_channel=PC)
        ocfg = ko.default_config(rotation_divisor=PD, rotation_channel=PC)
        try:
            cf = N.CorrelationFlow(cfg, H, W, max_batch=2, max_frames=6)
        except N.NikError as e:                      # (a polar map that leaves a very elongated image by more than one pixel is refused)
            assert e.code == N.NIK_ERR_UNSUPPORTED_SIZE, str(e)
            continue
        orc = ko.Oracle(ocfg, H, W)
        x = rng.random((W, H), dtype=np.float32)
        assert _relmax(cf.dbg_fft(x, 0), orc.fft(x)) < 3e-6, (H, W)
        assert np.array_equal(cf.dbg_polar(x), orc.polar(orc.fftshift(orc.remove_zero(x)))), (H, W, PD, PC)
        keys, curs, motions = synth.make_batch(2, H, W, seed0=900 + trial, max_shift=max(1, min(H, W) // 10), max_theta=6.0)
        for i in range(2):
            cf.intermedium_u8(keys[i], i)
            cf.intermedium_u8(curs[i], 2 + i)
        assert np.array_equal(cf.dbg_rotate(2, -11), orc.rotate(orc.normalize_u8(curs[0]), -5.5)), (H, W)
        for small in (True, False):
            res = cf.pose_batch([0, 1], [2, 3], small)
            poses, infos, dbgs, _ = ko.track_pairs(ocfg, keys, curs, small, nthreads=2)
            for i in range(2):
                def rerun(row, col, i=i):
                    o = ko.Oracle(ocfg, H, W); o.force_rotation(row, col)
                    kf, kp = o.intermedium(o.normalize_u8(keys[i])); ci = o.normalize_u8(curs[i]); _, cp = o.intermedium(ci)
                    return o.compute_pose(kf, ci, kp, cp, small)
                # tiny planes have shallow peaks (PSR ~ 10-30): the PSR tolerance of the full-size tests, the arg-max rule unchanged
                ok, _, msg = check_pose_parity(res[i], poses[i], infos[i], dbgs[i], PD, psr_rtol=5e-3, rerun=rerun)
                assert ok, "%dx%d polar %dx%d pair %d %s: %s" % (W, H, PD, PC, i, motions[i], msg)
        cf.close()
        done += 1
    assert done >= 8
