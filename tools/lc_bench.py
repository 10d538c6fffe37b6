import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch, synth
from kcc_helpers import nik
N = nik()
H, W = 480, 640
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
MB = 128
cf = N.CorrelationFlow(N.default_config(), H, W, max_batch=MB, max_frames=NC + 1)
U = 32
cv = [synth.canvas(900 + i, H, W) for i in range(4)]
uniq = np.stack([synth.window(cv[i % 4], H, W, (7 * i) % 60 - 30, (5 * i) % 80 - 40, 0.5 * (i % 11)) for i in range(U)])
d = torch.from_numpy(np.tile(uniq, (MB // U, 1, 1))).cuda(); torch.cuda.synchronize()
for b in range(0, NC, MB):
    m = min(MB, NC - b)
    cf.intermedium_batch_dev(d.data_ptr(), m, list(range(b, b + m)))
q = synth.window(cv[1], H, W, 3, -4, 0.0)
cf.intermedium_u8(q, NC)
cf.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); best, res, br = cf.match(NC, list(range(NC))); dt = time.perf_counter() - t0
print(json.dumps({"candidates": NC, "seconds": round(dt, 4), "candidates_per_s": round(NC / dt, 1), "best": best, "pose": br["pose"], "info": [round(v, 1) for v in br["info"]]}))
t0 = time.perf_counter(); b2, r2, short = cf.match_topk(NC, list(range(NC)), 16); dt2 = time.perf_counter() - t0
print(json.dumps({"topk16_seconds": round(dt2, 4), "candidates_per_s": round(NC / dt2, 1), "same_best_score": abs(sum(r2["info"]) - sum(br["info"])) < 1e-9}))
