"""pairs/s of one geometry through the batched device entry point (which kernel family runs: nik_is_generic mask).
usage: python tools/size_rate.py H W PD PC [batch]"""
import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, synth
from kcc_helpers import nik
N = nik(); H, W, PD, PC = map(int, sys.argv[1:5]); B = int(sys.argv[5]) if len(sys.argv) > 5 else 128
cf = N.CorrelationFlow(N.default_config(rotation_divisor=PD, rotation_channel=PC), H, W, max_batch=B, max_frames=2 * B)
k, c, _ = synth.make_batch(16, H, W, seed0=3); k = np.tile(k, (B // 16, 1, 1)); c = np.tile(c, (B // 16, 1, 1))
dk, dc = torch.from_numpy(k).cuda(), torch.from_numpy(c).cuda(); torch.cuda.synchronize()
cf.intermedium_batch_dev(dk.data_ptr(), B, list(range(B)))
for _ in range(3): cf.track_batch_dev(dc.data_ptr(), list(range(B)), list(range(B, 2 * B)), True, sync=True)
# (asynchronous calls finalise their results into the caller's array up to two calls later: the arrays must outlive them)
ring = [(N.NikPoseResult * B)() for _ in range(3)]
t0 = time.perf_counter(); n = 12
for i in range(n): cf.track_batch_dev(dc.data_ptr(), list(range(B)), list(range(B, 2 * B)), True, sync=False, res=ring[i % 3])
cf.synchronize(); dt = (time.perf_counter() - t0) / n
print("%dx%d polar %dx%d: family mask %d, %.0f pairs/s" % (W, H, PD, PC, cf._L.nik_is_generic(cf._ctx), B / dt), flush=True)
