"""Latency of small synchronous batches (configs[1] at B = 1..64): separate launches vs hipGraph replay (nik_set_graphs).
Prints one JSON line; bench.py --workload sequence reports the tracker with both."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, synth
from kcc_helpers import nik
N = nik(); H, W = 480, 640
cf = N.CorrelationFlow(N.default_config(), H, W, max_batch=64, max_frames=130)
keys, curs, _ = synth.make_unique_batch(64, H, W, seed0=5, max_theta=5.0)
import torch
dk = torch.from_numpy(keys).cuda(); dc = torch.from_numpy(curs).cuda(); torch.cuda.synchronize()
cf.intermedium_batch_dev(dk.data_ptr(), 64, list(range(64))); cf.intermedium_batch_dev(dc.data_ptr(), 64, list(range(64, 128))); cf.synchronize()
out = {}
for cache in (0, 1):
    cf.set_kzz_cache(bool(cache))
    for g in (0, 64):
        cf.set_graphs(g)
        for B in (1, 2, 4, 8, 16, 32, 64):
            ks, cs = list(range(B)), list(range(64, 64 + B))
            for _ in range(5): cf.pose_batch(ks, cs, True)
            t0 = time.perf_counter()
            for _ in range(200): cf.pose_batch(ks, cs, True)
            out["kzz%d_graph%d_B%d_ms" % (cache, 1 if g else 0, B)] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
print(json.dumps(out))
