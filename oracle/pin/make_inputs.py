"""Seeded inputs for oracle/pin/pin_reference (run from the repository root): the same tests/synth.py pairs the parity
tests use, written as raw u8 files plus a manifest the C++ program reads.  Regenerated from seeds on the consuming side
(tests/test_ref_golden.py), so only the reference's OUTPUTS travel."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

CASES = [  # name, H, W, PD, PC, n pairs, seed0, max_theta
    ("small", 60, 80, 120, 80, 8, 5000, 10.0),
    ("full", 480, 640, 720, 480, 8, 5100, 10.0),
    ("full_large_rot", 480, 640, 720, 480, 4, 5200, 80.0),
]


def pairs_of(case):
    name, H, W, PD, PC, n, seed0, mt = case
    small = dict(max_shift=max(1, min(H, W) // 10), base_shift=max(1, min(H, W) // 8)) if H < 200 else {}
    return synth.make_unique_batch(n, H, W, seed0=seed0, max_theta=mt, **small)


if __name__ == "__main__":
    out = os.path.join(ROOT, "oracle", "pin", "_work")
    os.makedirs(out, exist_ok=True)
    man = []
    for case in CASES:
        name, H, W, PD, PC, n, seed0, mt = case
        keys, curs, motions = pairs_of(case)
        keys.tofile(os.path.join(out, name + "_keys.u8")); curs.tofile(os.path.join(out, name + "_curs.u8"))
        man.append(dict(name=name, H=H, W=W, PD=PD, PC=PC, n=n, seed0=seed0, max_theta=mt, not_large_rotation=int(mt <= 10.0),
                        motions=[list(m) for m in motions]))
    # (a flat text manifest as well: the C++ side needs no JSON parser)
    with open(os.path.join(out, "manifest.txt"), "w") as f:
        for m in man:
            f.write("%s %d %d %d %d %d %d\n" % (m["name"], m["H"], m["W"], m["PD"], m["PC"], m["n"], m["not_large_rotation"]))
    json.dump(man, open(os.path.join(out, "manifest.json"), "w"), indent=1)
    print("wrote", out)
