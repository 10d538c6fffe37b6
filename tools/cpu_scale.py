"""CPU oracle thread scaling on the GPU box's host (tool): pairs/s of ora_track_pairs by OpenMP thread count, second (warm) run
of each count -- the first run of a thread count pays the page faults of its fresh per-thread heaps."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, synth
from oracle import kcc_oracle as ko
cfg = ko.default_config()
k, c, _ = synth.make_batch(32, 480, 640, seed0=1, max_shift=48)
k = np.tile(k, (32, 1, 1)); c = np.tile(c, (32, 1, 1))
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for nt in (1, 8, 16, 32, 64, 128, 256):
    n = min(1024, max(8, nt * 4))
    _, _, _, s0 = ko.track_pairs(cfg, k[:n], c[:n], True, nthreads=nt)
    _, _, _, s = ko.track_pairs(cfg, k[:n], c[:n], True, nthreads=nt)
    print(nt, "threads", n, "pairs: cold %.1f, warm %.1f pairs/s" % (n / s0, n / s), flush=True)
