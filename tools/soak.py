"""Soak test (tool): context create/destroy churn must return device memory; 3000 back-to-back steps of 256 pairs must keep
producing identical results at the sustained rate."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch, synth
from kcc_helpers import nik, SMALL
N = nik()
# 1. create/destroy churn: device memory must come back
free0 = torch.cuda.mem_get_info()[0]
for i in range(20):
    cf = N.CorrelationFlow(N.default_config(), 480, 640, max_batch=64, max_frames=128)
    cf.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("create/destroy x20: free before %.2f GB after %.2f GB" % (free0 / 1e9, free1 / 1e9))
# 2. sustained load: 3000 steps of 256 pairs, results must stay identical
cf = N.CorrelationFlow(N.default_config(), 480, 640, max_batch=256, max_frames=512)
keys, curs, _ = synth.make_batch(32, 480, 640, seed0=1, max_shift=48, max_theta=10.0)
dk = torch.from_numpy(np.tile(keys, (8, 1, 1))).cuda(); dc = torch.from_numpy(np.tile(curs, (8, 1, 1))).cuda(); torch.cuda.synchronize()
cf.intermedium_batch_dev(dk.data_ptr(), 256, list(range(256)))
ref = [r.as_dict() for r in cf.track_batch_dev(dc.data_ptr(), list(range(256)), list(range(256, 512)), True, sync=True)]
ring = [(N.NikPoseResult * 256)() for _ in range(3)]
t0 = time.perf_counter(); bad = 0
for k in range(3000):
    cf.track_batch_dev(dc.data_ptr(), list(range(256)), list(range(256, 512)), True, sync=False, res=ring[k % 3])
    if k % 500 == 499:
        cf.synchronize()
        got = [r.as_dict() for r in ring[k % 3]]
        bad += sum(g != r for g, r in zip(got, ref))
cf.synchronize()
dt = time.perf_counter() - t0
print("3000 steps: %.1f k pairs/s, mismatching results: %d, free now %.2f GB" % (3000 * 256 / dt / 1e3, bad, torch.cuda.mem_get_info()[0] / 1e9))
