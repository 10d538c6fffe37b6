"""Generates tests/golden/camera_golden.json from the INDEPENDENT numpy restatement (oracle/np_restatement.py) of
the camera undistortion step (reference src/camera.cc:45-47, 92-93) -- never from the C oracle, the product host code
or the HIP path it is used to check.  "Parity unpinned" (no OpenCV here): see DESIGN.md.

Run from the repo root:  python tests/golden/make_camera_golden.py
Fixtures are small: new_K, CRC32 of the two maps and of one remapped synthetic frame, plus a few sampled entries.
"""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from oracle import np_restatement as npr  # noqa: E402

# (name, W, H, K = fx cx fy cy, D = k1 k2 p1 p2 k3)
CAMERAS = [
    ("vga_barrel", 640, 480, (420.0, 318.5, 418.0, 242.3), (-0.31, 0.12, 0.0008, -0.0005, -0.02)),
    ("vga_mild", 640, 480, (610.2, 322.8, 609.1, 236.4), (-0.08, 0.03, 0.0, 0.0, 0.0)),
    ("vga_pincushion", 640, 480, (500.0, 320.0, 500.0, 240.0), (0.12, -0.02, -0.001, 0.0012, 0.0)),
    ("vga_none", 640, 480, (600.0, 320.0, 600.0, 240.0), (0.0, 0.0, 0.0, 0.0, 0.0)),
    ("small_barrel", 80, 60, (52.0, 39.6, 51.5, 30.2), (-0.28, 0.09, 0.001, -0.0007, 0.0)),
    ("hd_barrel", 1280, 720, (830.0, 642.0, 828.0, 357.0), (-0.22, 0.06, 0.0004, 0.0003, -0.005)),
]


def frame(seed, H, W):
    return synth.window(synth.canvas(seed, H, W), H, W, 0, 0)


def case(name, W, H, K, D):
    newK = npr.optimal_new_camera_matrix(K, D, W, H)
    m1, m2 = npr.undistort_maps(K, D, newK, W, H)
    img = frame(77, H, W)
    und = npr.remap_u8(img, m1, m2)
    pts = [(0, 0), (H - 1, W - 1), (H // 2, W // 2), (H // 3, (2 * W) // 3), (H - 1, 0)]
    return dict(name=name, W=W, H=H, K=list(K), D=list(D), new_K=[float(v) for v in newK],
                map1_crc=zlib.crc32(m1.tobytes()), map2_crc=zlib.crc32(m2.tobytes()),
                frame_seed=77, undistorted_crc=zlib.crc32(und.tobytes()),
                samples=[dict(r=r, c=c, sx=int(m1[r, c, 0]), sy=int(m1[r, c, 1]), frac=int(m2[r, c]), px=int(und[r, c])) for r, c in pts])


def main():
    out = dict(cameras=[case(*c) for c in CAMERAS])
    path = os.path.join(ROOT, "tests", "golden", "camera_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
