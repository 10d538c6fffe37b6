import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import importlib, numpy as np, torch
N = importlib.import_module("ni-slam_amd.nislam_kcc"); import synth
H, W, B, LEVELS, R = 480, 640, 32, 4, 4
pyr = N.Pyramid(N.default_config(), H, W, levels=LEVELS, max_batch=B, device=0)
keys, curs, _ = synth.make_unique_batch(B, H, W, seed0=50, max_theta=8.0, max_shift=40)
dev = torch.device("cuda:0"); dk = torch.from_numpy(keys).to(dev); dc = torch.from_numpy(curs).to(dev)
ring = [(N.NikPoseResult * (LEVELS * B))() for _ in range(8)]
for k in range(5): pyr.track_dev_async(dk.data_ptr(), dc.data_ptr(), B, R, res=ring[k % 8])
pyr.synchronize()
for steps in (20, 60):
    t0 = time.perf_counter()
    for k in range(steps): pyr.track_dev_async(dk.data_ptr(), dc.data_ptr(), B, R, res=ring[k % 8])
    t1 = time.perf_counter(); pyr.synchronize(); t2 = time.perf_counter()
    print("steps", steps, "enqueue ms/batch", 1e3*(t1-t0)/steps, "total ms/batch", 1e3*(t2-t0)/steps, "pairs/s", B*steps/(t2-t0))
# one batch alone, synchronous
t0 = time.perf_counter()
for k in range(20): pyr.track_dev(dk.data_ptr(), dc.data_ptr(), B, R)
print("sync ms/batch", 1e3*(time.perf_counter()-t0)/20)
pyr.synchronize()
for rep in range(3):
    ts = []
    for k in range(6):
        t0 = time.perf_counter(); pyr.track_dev_async(dk.data_ptr(), dc.data_ptr(), B, R, res=ring[k % 8]); ts.append(1e3*(time.perf_counter()-t0))
    t0 = time.perf_counter(); pyr.synchronize(); tsync = 1e3*(time.perf_counter()-t0)
    print("per-call enqueue ms after a sync:", ["%.3f" % t for t in ts], "final sync", "%.3f" % tsync)
