"""placement_probe.py -- does the per-process spread of the HBM-bound kernels (kB<480,fwd_mul_inv>: 0.567 ... 0.625 ms per 512 pairs
between three runs of the same build on one box) come with the ALLOCATION?  One process creates the context R times (its buffers are
freed and allocated again each time) and times the per-kernel table each time; run the script P times for the process-to-process
spread.  usage: python tools/placement_probe.py [R]"""
import json, os, sys, time
R0 = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
import numpy as np, torch, synth
from kcc_helpers import nik
N = nik(); H, W, B = 480, 640, 512
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
keys, curs, _ = synth.make_unique_batch(64, H, W, seed0=1000)
dk = torch.from_numpy(np.tile(keys, (B // 64, 1, 1))).cuda(); dc = torch.from_numpy(np.tile(curs, (B // 64, 1, 1))).cuda(); torch.cuda.synchronize()
names = ["kB<480,fwd_mul_inv>", "kB<640,fwd_mul_inv>", "kA_inv<360,kernel_fwd>", "kB<640,fwd_abs_inv>", "kA_fwd<240,rot8>"]
for r in range(R):
    cf = N.CorrelationFlow(N.default_config(), H, W, max_batch=B, max_frames=2 * B)
    cf.set_streams(2)
    ks, cs = list(range(B)), list(range(B, 2 * B))
    cf.intermedium_batch_dev(dk.data_ptr(), B, ks); cf.synchronize()
    ring = [(N.NikPoseResult * B)() for _ in range(3)]
    for i in range(5): cf.track_batch_dev(dc.data_ptr(), ks, cs, True, sync=False, res=ring[i % 3])
    cf.synchronize(); t0 = time.perf_counter()
    for i in range(20): cf.track_batch_dev(dc.data_ptr(), ks, cs, True, sync=False, res=ring[i % 3])
    cf.synchronize(); rate = B * 20 / (time.perf_counter() - t0)
    cf.set_streams(1); cf.profile_enable(True)
    for i in range(10): cf.track_batch_dev(dc.data_ptr(), ks, cs, True, sync=False, res=ring[i % 3])
    cf.synchronize()
    st = {s["name"]: s["ms"] / max(s["launches"], 1) for s in cf.profile_read()}
    cf.profile_enable(False); cf.close()
    print("pid %d create %d: %.1f pairs/s  " % (os.getpid(), r, rate) + "  ".join("%s %.4f" % (n.split("<")[1][:14], st[n]) for n in names) + "  sum %.3f" % sum(st.values()), flush=True)
