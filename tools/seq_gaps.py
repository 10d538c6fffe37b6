"""key-frame spacing of bench.py's synthetic sequences (what the tracker's chain speculation has to guess)"""
import sys, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import importlib, numpy as np, torch
N = importlib.import_module("ni-slam_amd.nislam_kcc"); import synth
H, W, T = 480, 640, 512
cv = synth.canvas(4242, H, W)
base = [synth.window(cv, H, W, int(3 * i) % 200 - 100, int(2 * i) % 160 - 80, 0.5 * (i % 9)) for i in range(64)]
seq = np.stack([base[i % 64] for i in range(T)]); d = torch.from_numpy(seq).to("cuda:0")
flow = N.CorrelationFlow(N.default_config(), H, W, max_batch=64, max_frames=T + 70); flow.set_kzz_cache(True)
trk = N.Tracker(flow, N.tracker_config())
outs = []
for b0 in range(0, T, 64): outs += trk.push_dev(d[b0:b0 + 64].data_ptr(), 64)
keys = [o["frame_id"] for o in outs if o["inserted"]]
gaps = [b - a for a, b in zip(keys, keys[1:])]
print("gaps:", gaps[:120])
print(collections.Counter(gaps))
