import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, synth
from kcc_helpers import nik
N = nik(); H, W, B = 480, 640, 32
keys, curs, _ = synth.make_unique_batch(B, H, W, seed0=50, max_theta=8.0, max_shift=40)
dk = torch.from_numpy(keys).cuda(); dc = torch.from_numpy(curs).cuda(); torch.cuda.synchronize()
for levels in (4, 3, 2, 1):
    pyr = N.Pyramid(N.default_config(), H, W, levels=levels, max_batch=B, device=0)
    ring = [(N.NikPoseResult * (levels * B))() for _ in range(3)]
    for k in range(6): pyr.track_dev_async(dk.data_ptr(), dc.data_ptr(), B, 4, res=ring[k % 3])
    pyr.synchronize()
    t0 = time.perf_counter(); n = 200
    for k in range(n): pyr.track_dev_async(dk.data_ptr(), dc.data_ptr(), B, 4, res=ring[k % 3])
    pyr.synchronize(); dt = (time.perf_counter() - t0) / n
    print("levels", levels, "ms per batch %.4f" % (1e3 * dt), "pairs/s %.0f" % (B / dt), flush=True)
    pyr.close()
