import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch, synth
from kcc_helpers import nik, check_pose_parity
from oracle import kcc_oracle as ko
N = nik(); H, W, PD = 480, 640, 720
B = 256
keys, curs, mot = synth.make_unique_batch(B, H, W, seed0=1000, max_theta=10.0)
cf = N.CorrelationFlow(N.default_config(), H, W, max_batch=B, max_frames=2 * B)
dk = torch.from_numpy(keys).cuda(); dc = torch.from_numpy(curs).cuda(); torch.cuda.synchronize()
cf.intermedium_batch_dev(dk.data_ptr(), B, list(range(B)))
res = [r.as_dict() for r in cf.track_batch_dev(dc.data_ptr(), list(range(B)), list(range(B, 2 * B)), True, sync=True)]
poses, infos, dbgs, _ = ko.track_pairs(ko.default_config(), keys, curs, True, nthreads=32)
bad = 0
for i in range(B):
    ok, ex, msg = check_pose_parity(res[i], poses[i], infos[i], dbgs[i], PD)
    if not ok:
        bad += 1; print(i, mot[i], msg)
print("bad", bad)
# sequence
cv = synth.canvas(4242, H, W)
base = [synth.window(cv, H, W, int(3 * i) % 200 - 100, int(2 * i) % 160 - 80, 0.5 * (i % 9)) for i in range(64)]
T = 96
seq = np.stack([base[i % 64] for i in range(T)]); d_seq = torch.from_numpy(seq).cuda()
def run(win):
    flow = N.CorrelationFlow(N.default_config(), H, W, max_batch=win, max_frames=T + win + 2); flow.set_kzz_cache(True)
    trk = N.Tracker(flow, N.tracker_config()); outs = []
    for b0 in range(0, T, win):
        m = min(win, T - b0); outs += trk.push_dev(d_seq[b0:b0 + m].data_ptr(), m)
    trk.close(); flow.close(); return outs
a, b = run(64), run(1)
n = 0
for i in range(T):
    if a[i] != b[i]:
        n += 1
        if n < 6: print(i, {k: (a[i][k], b[i][k]) for k in a[i] if a[i][k] != b[i][k]})
print("seq mismatches", n)
