"""Experiment: two independent pyramids fed alternately -- does throughput rise when the level chains have more slack?"""
import sys, time, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import importlib, numpy as np, torch
N = importlib.import_module("ni-slam_amd.nislam_kcc"); import synth
H, W, B, LEVELS, Rr = 480, 640, 32, 4, 4
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pyrs = [N.Pyramid(N.default_config(), H, W, levels=LEVELS, max_batch=B, device=0) for _ in range(NP)]
keys, curs, _ = synth.make_unique_batch(B, H, W, seed0=50, max_theta=8.0, max_shift=40)
dev = torch.device("cuda:0"); dk = torch.from_numpy(keys).to(dev); dc = torch.from_numpy(curs).to(dev)
rings = [[(N.NikPoseResult * (LEVELS * B))() for _ in range(4)] for _ in range(NP)]
def go(steps):
    for k in range(steps): pyrs[k % NP].track_dev_async(dk.data_ptr(), dc.data_ptr(), B, Rr, res=rings[k % NP][(k // NP) % 4])
    for p in pyrs: p.synchronize()
go(8)
for steps in (40, 120):
    t0 = time.perf_counter(); go(steps); dt = time.perf_counter() - t0
    print("pyramids", NP, "steps", steps, "ms/batch", 1e3 * dt / steps, "pairs/s", B * steps / dt)
