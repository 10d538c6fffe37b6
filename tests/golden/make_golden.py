"""Generates tests/golden/kcc_golden.json from the INDEPENDENT numpy/scipy restatement
(oracle/np_restatement.py) -- never from the C oracle or the HIP path it is used to check.

Run from the repo root:  python tests/golden/make_golden.py
The reference itself cannot produce vectors here (unbuildable: FFTW3f/Eigen3/OpenCV absent), so these
fixtures pin the C oracle against a second implementation, not against the reference binary
("parity unpinned").
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from oracle import np_restatement as npr  # noqa: E402

GEOMS = {"small": dict(H=60, W=80, PD=120, PC=80), "full": dict(H=480, W=640, PD=720, PC=480)}


def pose_case(geom_name, seed, dy, dx, theta, kernel, small_rot):
    g = GEOMS[geom_name]
    cf = npr.CorrelationFlowNp(g["H"], g["W"], g["PD"], g["PC"], kernel=kernel)
    key, cur = synth.make_pair(seed, g["H"], g["W"], dy, dx, theta)
    ki = (key.T.astype(np.float32) / np.float32(255.0))
    ci = (cur.T.astype(np.float32) / np.float32(255.0))
    kf, kp = cf.intermedium(ki)
    cf_, cp = cf.intermedium(ci)
    pose, info, dbg = cf.compute_pose(kf, ci, kp, cp, small_rot)
    _, _, rr, rc, grot = cf.estimate_trans(kp, cp, 1)
    peak = float(grot[rc, rr])
    mirror = float(grot[rc, (rr + g["PD"] // 2) % g["PD"]])
    return dict(geom=geom_name, seed=seed, dy=dy, dx=dx, theta=theta, kernel=kernel, small_rot=bool(small_rot),
                pose=[float(v) for v in pose], info=[float(v) for v in info],
                rot_row=int(dbg["rot_row"]), rot_col=int(dbg["rot_col"]),
                trans_row=[int(v) for v in dbg["trans_row"]], trans_col=[int(v) for v in dbg["trans_col"]],
                chosen=int(dbg["chosen"]), n_hyp=int(dbg["n_hyp"]),
                rot_gap=abs(peak - mirror) / abs(peak),
                spectrum_checksum=[float(np.abs(cf_).sum(dtype=np.float64)), float(np.abs(cp).sum(dtype=np.float64))])


def main():
    out = dict(geoms=GEOMS, pose_cases=[], fft_vectors=[], gather_cases=[])
    rng = np.random.default_rng(2024)
    for i in range(10):
        dy, dx = (int(v) for v in rng.integers(-6, 7, 2))
        th = float(np.round(rng.uniform(-12, 12), 3)) if i % 3 else 0.0
        out["pose_cases"].append(pose_case("small", 1000 + i, dy, dx, th, 0, i % 2 == 0))
    for i in range(3):
        dy, dx = (int(v) for v in rng.integers(-6, 7, 2))
        out["pose_cases"].append(pose_case("small", 1100 + i, dy, dx, 3.0 * i, 1, True))
    for i, (dy, dx, th, sr) in enumerate([(17, -23, 0.0, True), (-40, 31, 6.5, True), (8, 12, -9.25, False)]):
        out["pose_cases"].append(pose_case("full", 2000 + i, dy, dx, th, 0, sr))
    # a tiny FFT known-answer vector (rows=6, cols=4) through the reference's r2c layout
    x = rng.random((4, 6)).astype(np.float32)
    xf = npr.fft(x)
    out["fft_vectors"].append(dict(rows=6, cols=4, x=x.reshape(-1).tolist(),
                                   xf_re=xf.real.reshape(-1).tolist(), xf_im=xf.imag.reshape(-1).tolist()))
    # gathers: seeded plane -> polar / rotate, pinned by exact float checksums + samples
    g = GEOMS["small"]
    plane = np.random.default_rng(77).random((g["W"], g["H"]), dtype=np.float32)
    pol = npr.polar(npr.fftshift(npr.remove_zero(plane)), g["PD"], g["PC"])
    case = dict(geom="small", seed=77, polar_sum=float(pol.sum(dtype=np.float64)),
                polar_samples=[float(pol[j, i]) for j, i in [(0, 0), (5, 7), (40, 60), (79, 119), (79, 30)]], rotations=[])
    for deg in (0.0, 0.5, -13.5, 90.0, 180.0, 355.0):
        r = npr.rotate(plane, deg)
        case["rotations"].append(dict(degree=deg, sum=float(r.sum(dtype=np.float64)),
                                      samples=[float(r[c, rr]) for c, rr in [(0, 0), (3, 9), (40, 30), (79, 59)]]))
    out["gather_cases"].append(case)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kcc_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(out["pose_cases"]), "pose cases")


if __name__ == "__main__":
    main()
