"""Dry run of the CONSUMING side of the pinning kit, for an environment without the reference's libraries: writes stand-in
"reference" vectors produced by the oracle itself into a TEMPORARY directory, runs tests/test_ref_golden.py against them
(NIK_REF_GOLDEN_DIR) and deletes them.  It proves the file formats, array conventions and comparisons of the tests are
consistent -- it pins nothing (the oracle is compared with itself).  Run from the repository root."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle", "pin")):
    sys.path.insert(0, p)
from oracle import kcc_oracle as O  # noqa: E402
import make_inputs  # noqa: E402
import test_ref_golden as t  # noqa: E402

with tempfile.TemporaryDirectory() as G:
    H, W, PD, PC = 60, 80, 120, 80
    x, z = t.lcg_floats(H, W, 12345), t.lcg_floats(H, W, 777)
    ora = O.Oracle(O.default_config(rotation_divisor=PD, rotation_channel=PC), H, W)
    w = lambda name, a: a.tofile(os.path.join(G, name))      # noqa: E731
    xf = ora.fft(x)
    w("ref_polar.bin", ora.polar(x)); w("ref_fft.bin", xf); w("ref_ifft.bin", ora.ifft(xf))
    bad = xf.copy(); bad[:, 0] += 3.5j; bad[:, H // 2] += -2.25j
    w("ref_ifft_nonhermitian.bin", ora.ifft(bad))
    degs = [0.5, 37, 180, -12.5, 90]
    for k, d in enumerate(degs):
        w("ref_rotate_%d.bin" % k, O.Oracle.rotate(x, float(d)))
    K, D = [52.0, 39.6, 51.5, 30.2], [-0.28, 0.09, 0.001, -0.0007, 0.0]
    newK = O.optimal_new_camera_matrix(K, D, W, H)
    m1, m2 = O.undistort_maps(K, D, newK, W, H)
    w("ref_map1.bin", m1); w("ref_map2.bin", m2)
    w("ref_remap_u8.bin", O.remap_u8(t.lcg_bytes(H * W, 999).reshape(H, W), m1, m2))
    rgb = t.lcg_bytes(3 * H * W, 4242).reshape(H, W, 3).astype(np.int64)
    w("ref_rgb2gray.bin", ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8))
    b = (x + np.float32(0.1)).astype(np.float64)
    w("ref_pow3.bin", (b * b * b).astype(np.float32))
    psr, tr = ora.estimate_trans(ora.fft(z), xf, 0)[:2]
    json.dump(dict(opencv="STAND-IN (oracle)", H=H, W=W, PD=PD, PC=PC, estimate_trans=dict(trans=list(tr), psr=psr), rotate_degrees=degs,
                   camera=dict(K=K, D=D, new_K=[float(v) for v in np.ravel(newK)]), maxcoeff_tie=dict(set=[[7, 3], [2, 9], [40, 3]], row=7, col=3)),
              open(os.path.join(G, "ref_recalled.json"), "w"))
    cases = []
    for case in make_inputs.CASES[:2]:
        name, H2, W2, PD2, PC2, n, seed0, mt = case
        keys, curs, _ = make_inputs.pairs_of(case)
        ocfg = O.default_config(rotation_divisor=PD2, rotation_channel=PC2)
        poses, infos, _, _ = O.track_pairs(ocfg, keys, curs, mt <= 10.0, nthreads=4)
        o2 = O.Oracle(ocfg, H2, W2)
        prs = []
        for i in range(n):
            kf, kp = o2.intermedium(o2.normalize_u8(keys[i]))
            prs.append(dict(pose=list(poses[i]), info=list(infos[i]),
                            F_probe=[float(kf[0, 0].real), float(kf[2, 1].real), float(kf[2, 1].imag), float(kf[W2 - 1, H2 // 2].real)],
                            P_probe=[float(kp[0, 0].real), float(kp[2, 1].real), float(kp[2, 1].imag), float(kp[PC2 - 1, PD2 // 2].real)],
                            F_abs_sum=float(np.abs(kf).sum()), P_abs_sum=float(np.abs(kp).sum())))
        cases.append(dict(name=name, H=H2, W=W2, PD=PD2, PC=PC2, not_large_rotation=int(mt <= 10.0), pairs=prs))
    json.dump(dict(note="STAND-IN (oracle)", cases=cases), open(os.path.join(G, "ref_pairs.json"), "w"))
    args = sys.argv[1:] or ["-m", "not gpu"]
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_ref_golden.py"), "-q"] + args,
                       env=dict(os.environ, NIK_REF_GOLDEN_DIR=G), cwd=ROOT)
    sys.exit(r.returncode)
