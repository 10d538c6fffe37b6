"""Shared helpers of the parity tests."""
import importlib.util
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ni-slam_amd")

SMALL = dict(H=60, W=80, PD=120, PC=80)          # quick geometry (every FFT length instantiated)
FULL = dict(H=480, W=640, PD=720, PC=480)        # reference configs/config_ntu.yaml


def load_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_nik = None


def nik():
    """the ctypes binding of the HIP library (ni-slam_amd/nislam_kcc.py; the directory name is not importable)"""
    global _nik
    if _nik is None:
        _nik = load_module("nislam_kcc", os.path.join(PKG, "nislam_kcc.py"))
    return _nik


def ang_diff(a, b):
    d = (a - b) % (2 * math.pi)
    return min(d, 2 * math.pi - d)


# relative gap between the rotation arg-max and its 180-degree mirror below which the two peaks are a
# rounding-noise tie (the polar source is point-symmetric; see DESIGN.md "The 180-degree rotation tie").
# MEASURED, not guessed (tests/test_gpu_wide.py::test_response_noise_bounds_the_tie_tolerance, profiles/r03_response_noise.json):
# from identical float32 spectra the rotation response of the CPU float32 oracle deviates from the float64 evaluation by
# up to 6.0e-4 of the peak and the HIP path by up to 5.8e-4 (EstimateTrans divides by Kzz + lambda, whose small bins carry
# percent-level float32 error); two peaks closer than the sum of those deviations (1.2e-3) can legitimately swap.  The
# constant IS that measured sum (the largest gap ever accepted in the 512 + 6144 + 2 x 3072 pair sweeps was 9.2e-4); the
# test keeps it between 1x and 4x of what it measures.
ROT_TIE_REL = 1.2e-3
# The gaussian kernel multiplies the rounding error of xz by 2 / sigma^2 (= 50 at the reference's sigma 0.2) inside exp():
# both float32 implementations sit correspondingly further from the float64 response, measured the same way
# (tests/test_gpu_wide.py::test_response_noise_gaussian_kernel, profiles/r03_response_noise_gaussian.json).
ROT_TIE_REL_GAUSS = 4.0e-3


def check_pose_parity(gpu, ora_pose, ora_info, ora_dbg, PD, psr_rtol=2e-3, rerun=None, tie_rel=None):
    """gpu: dict from NikPoseResult.as_dict(); ora_*: oracle outputs.  Returns (ok, exact_rot, message).

    Rule: translation arg-max indices bit-exact; rotation arg-max bit-exact unless the oracle's own two
    mirror peaks are within ROT_TIE_REL of each other, in which case row may differ by PD/2 (same rotation
    modulo 180 deg, decided by FFT rounding noise in the reference itself); theta equal modulo 2*pi;
    PSR within psr_rtol.

    rerun (optional): callable (row, col) -> (pose, info, dbg) that re-runs the oracle with the rotation arg-max
    IMPOSED at the GPU's position (Oracle.force_rotation).  It generalises the tie rule to any near-tie of the
    rotation surface (e.g. a true rotation halfway between two 0.5-degree bins): if the oracle's response at the
    GPU's position is within ROT_TIE_REL of the oracle's maximum, the GPU must match that imposed run exactly."""
    tie_rel = ROT_TIE_REL if tie_rel is None else tie_rel
    exact_rot = gpu["rot_row"] == ora_dbg["rot_row"] and gpu["rot_col"] == ora_dbg["rot_col"]
    if not exact_rot:
        gap = abs(ora_dbg["rot_peak"] - ora_dbg["rot_mirror"]) / max(abs(ora_dbg["rot_peak"]), 1e-30)
        mirror = gpu["rot_col"] == ora_dbg["rot_col"] and (gpu["rot_row"] - ora_dbg["rot_row"]) % PD == PD // 2
        if not (mirror and gap < tie_rel):
            if rerun is not None:
                pose2, info2, dbg2 = rerun(gpu["rot_row"], gpu["rot_col"])
                gap2 = (ora_dbg["rot_peak"] - dbg2["rot_peak"]) / max(abs(ora_dbg["rot_peak"]), 1e-30)
                if gap2 < tie_rel:
                    ok, _, msg = _compare(gpu, pose2, info2, psr_rtol)
                    return ok, False, ("near-tie gap=%.2e: " % gap2) + msg
            ok, _, msg = _compare(gpu, ora_pose, ora_info, psr_rtol)
            return False, False, "rot argmax gpu=(%d,%d) oracle=(%d,%d) gap=%.2e; %s" % (
                gpu["rot_row"], gpu["rot_col"], ora_dbg["rot_row"], ora_dbg["rot_col"], gap, msg)
    ok, _, msg = _compare(gpu, ora_pose, ora_info, psr_rtol)
    return ok, exact_rot, msg


def parity_detail(gpu, ora_pose, ora_info, ora_dbg, PD, psr_rtol=2e-3, rerun=None, tie_rel=None):
    """check_pose_parity with the LETTER of the result spelled out.  Returns a dict:
      ok            the rule of check_pose_parity holds
      kind          "identical" (rotation arg-max row and column equal the oracle's), "mirror_tie" (row differs by PD/2 inside the
                    measured tie tolerance), "near_tie" (accepted through the imposed re-run) or "fail"
      theta_equal   the returned theta equals the oracle's value exactly (not merely modulo 2 pi)
      theta_2pi     theta differs from the oracle's by +-2 pi (to float precision: theta IS a float in the reference,
                    correlation_flow.cc:136): the |deg| > 90 -> deg - 180 fold of :108 applied to the mirror row leaves e.g.
                    -352 deg where the oracle's row gives +8 deg
    A pair whose rotation rows are identical always has theta_equal (same integer row -> same double arithmetic)."""
    ok, exact_rot, msg = check_pose_parity(gpu, ora_pose, ora_info, ora_dbg, PD, psr_rtol, rerun, tie_rel)
    if not ok:
        kind = "fail"
    elif exact_rot:
        kind = "identical"
    elif msg.startswith("near-tie"):
        kind = "near_tie"
    else:
        kind = "mirror_tie"
    d = gpu["pose"][2] - ora_pose[2]
    return dict(ok=ok, kind=kind, theta_equal=bool(d == 0.0), theta_2pi=bool(abs(abs(d) - 2 * math.pi) < 1e-5), message=msg)


def parity_summary(details):
    """counts over a list of parity_detail() results: what the bench line's parity_spot_check object and tools/parity_sweep.py report"""
    n = len(details)
    return dict(ok=bool(n > 0 and all(d["ok"] for d in details)), pairs=n,
                rotation_rows_identical=sum(d["kind"] == "identical" for d in details),
                accepted_mirror_ties=sum(d["kind"] == "mirror_tie" for d in details),
                accepted_near_ties=sum(d["kind"] == "near_tie" for d in details),
                failures=sum(d["kind"] == "fail" for d in details),
                theta_equal_to_oracle=sum(d["theta_equal"] for d in details),
                theta_differs_by_2pi=sum(d["theta_2pi"] for d in details),
                note="translation arg-max indices are compared exactly for every pair; theta_differs_by_2pi counts accepted ties whose theta is the oracle's +- 2 pi "
                     "(the deg - 180 fold of correlation_flow.cc:108 on the mirror row); every pair with identical rotation rows has theta equal to the oracle's exactly")


def _compare(gpu, ora_pose, ora_info, psr_rtol):
    msgs = []
    if gpu["pose"][0] != ora_pose[0] or gpu["pose"][1] != ora_pose[1]:
        msgs.append("translation gpu=%s oracle=%s" % (gpu["pose"][:2], list(ora_pose[:2])))
    if ang_diff(gpu["pose"][2], ora_pose[2]) > 1e-6:
        msgs.append("theta gpu=%r oracle=%r" % (gpu["pose"][2], ora_pose[2]))
    for k in (0, 2):
        if abs(gpu["info"][k] - ora_info[k]) > psr_rtol * abs(ora_info[k]):
            msgs.append("info[%d] gpu=%r oracle=%r" % (k, gpu["info"][k], ora_info[k]))
    return (not msgs), True, "; ".join(msgs)


def imposed_rerun(ocfg, H, W, key_img, cur_img, not_large_rotation):
    """rerun callable for check_pose_parity: the oracle's ComputePose for one u8 pair with the rotation arg-max imposed"""
    def rerun(row, col):
        from oracle import kcc_oracle as ko
        o = ko.Oracle(ocfg, H, W)
        o.force_rotation(row, col)
        kf, kp = o.intermedium(o.normalize_u8(key_img))
        x = o.normalize_u8(cur_img)
        _, xp = o.intermedium(x)
        return o.compute_pose(kf, x, kp, xp, not_large_rotation)
    return rerun


def tuning_lib():
    """path of the tuning library (ni-slam_amd/build.py build_tuning(): -DKCC_ABLATE, every laboratory switch of kcc_tune.h alive)"""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return os.path.join(root, "ni-slam_amd", "libnislam_kcc_hip_tune.so")


def run_under_tuning_lib(pytest_args, extra_env=None, timeout=850):
    """Run pytest in a subprocess whose binding loads the TUNING library ($NIK_LIB): the release library has no laboratory
    switches, so tests of a non-default form ($NIK_RING, $NIK_FUSE_FIX_ZERO=0 ...) run there.  The tuning library is built by
    __graft_entry__.build(); its absence is a failure, not a skip."""
    import os
    import subprocess
    import sys
    lib = tuning_lib()
    assert os.path.exists(lib), "tuning library missing: run __graft_entry__.build() (ni-slam_amd/build.py --tune)"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NIK_LIB=lib, NIK_UNDER_TUNING_LIB="1", **(extra_env or {}))
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x"] + list(pytest_args), cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout and "failed" not in p.stdout, p.stdout[-2000:]
    return p.stdout
