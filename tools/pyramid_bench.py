"""BASELINE config 3: 640x480 "stereo" (two independent mono streams) + 4-level pyramid with radius-4 lookup,
batch 32 frame pairs per step on one GPU.  Prints one JSON line (an extra measurement, not bench.py's headline)."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch, synth
from kcc_helpers import nik
N = nik()
H, W, B, LEVELS, R = 480, 640, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 4, 4
pyr = N.Pyramid(N.default_config(), H, W, levels=LEVELS, max_batch=B)
keys, curs, _ = synth.make_batch(min(B, 16), H, W, seed0=50, max_shift=40, max_theta=8.0)
reps = (B + len(keys) - 1) // len(keys)
dk = torch.from_numpy(np.tile(keys, (reps, 1, 1))[:B]).cuda(); dc = torch.from_numpy(np.tile(curs, (reps, 1, 1))[:B]).cuda()
torch.cuda.synchronize()
for _ in range(3):
    res = pyr.track_dev(dk.data_ptr(), dc.data_ptr(), B, R)
steps = 20
t0 = time.perf_counter()
for _ in range(steps):
    res = pyr.track_dev(dk.data_ptr(), dc.data_ptr(), B, R)
dt = (time.perf_counter() - t0) / steps
# bytes: key + current intermedium and one pose per level, level l has 1/4^l of the pixels (polar sizes 1, 4/9, 1/9, 1/36)
print(json.dumps({"config": "configs[2]: 640x480 stereo, 4-level pyramid, radius-4 lookup", "pairs_per_step": B, "ms_per_step": round(1e3 * dt, 3),
                  "pairs_per_s": round(B / dt, 1), "level0_pose_example": res[0][0]["pose"], "levels": pyr.dims}))
