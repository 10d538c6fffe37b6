"""B=1 latency of the hot path (BASELINE config 2): one frame registered against one keyframe, inputs in HBM,
synchronous call -> the time a real-time tracker would see per frame.  Also small batches."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch, synth
from kcc_helpers import nik
N = nik()
H, W = 480, 640
out = {}
for B in (1, 2, 4, 8, 16, 32):
    cf = N.CorrelationFlow(N.default_config(), H, W, max_batch=B, max_frames=2 * B)
    keys, curs, _ = synth.make_batch(B, H, W, seed0=7, max_shift=40, max_theta=8.0)
    dk = torch.from_numpy(keys).cuda(); dc = torch.from_numpy(curs).cuda(); torch.cuda.synchronize()
    cf.intermedium_batch_dev(dk.data_ptr(), B, list(range(B))); cf.synchronize()
    for cache in (False, True):
        cf.set_kzz_cache(cache)
        for _ in range(5):
            cf.track_batch_dev(dc.data_ptr(), list(range(B)), list(range(B, 2 * B)), True, sync=True)
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            cf.track_batch_dev(dc.data_ptr(), list(range(B)), list(range(B, 2 * B)), True, sync=True)
        dt = (time.perf_counter() - t0) / n
        out["B%d%s" % (B, "_kzz" if cache else "")] = {"ms_per_call": round(1e3 * dt, 4), "pairs_per_s": round(B / dt, 1)}
    cf.close()
print(json.dumps(out))
