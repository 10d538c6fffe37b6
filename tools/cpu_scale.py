import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, synth
from oracle import kcc_oracle as ko
cfg = ko.default_config()
k, c, _ = synth.make_batch(32, 480, 640, seed0=1, max_shift=48)
k = np.tile(k, (8, 1, 1)); c = np.tile(c, (8, 1, 1))
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for nt in (1, 8, 16, 32, 64, 128, 256):
    n = min(256, max(8, nt * 2))
    _, _, _, s = ko.track_pairs(cfg, k[:n], c[:n], True, nthreads=nt)
    print(nt, "threads", n, "pairs", round(n / s, 2), "pairs/s")
