"""Soak test of round 4's additions (tool): any-size contexts created / destroyed must return device memory; sustained load on the
any-size kernels and on a mixed context keeps producing identical results; nik_tracker_push_host over many windows from pinned and
pageable memory keeps agreeing with push_dev and leaves no upload in flight."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch, synth
from kcc_helpers import nik
N = nik()
free0 = torch.cuda.mem_get_info()[0]
for i in range(12):
    for (H, W, PD, PC) in ((480, 752, 720, 480), (480, 640, 720, 64), (62, 94, 90, 50)):
        cf = N.CorrelationFlow(N.default_config(rotation_divisor=PD, rotation_channel=PC), H, W, max_batch=64, max_frames=128)
        cf.close()
torch.cuda.synchronize()
print("any-size create/destroy x36: free before %.2f GB after %.2f GB" % (free0 / 1e9, torch.cuda.mem_get_info()[0] / 1e9), flush=True)
for (H, W, PD, PC, steps) in ((480, 752, 720, 480, 150), (480, 640, 720, 64, 1500)):
    B = 128
    cf = N.CorrelationFlow(N.default_config(rotation_divisor=PD, rotation_channel=PC), H, W, max_batch=B, max_frames=2 * B)
    keys, curs, _ = synth.make_batch(16, H, W, seed0=1, max_shift=30, max_theta=8.0)
    dk = torch.from_numpy(np.tile(keys, (8, 1, 1))).cuda(); dc = torch.from_numpy(np.tile(curs, (8, 1, 1))).cuda(); torch.cuda.synchronize()
    cf.intermedium_batch_dev(dk.data_ptr(), B, list(range(B)))
    ref = [r.as_dict() for r in cf.track_batch_dev(dc.data_ptr(), list(range(B)), list(range(B, 2 * B)), True, sync=True)]
    ring = [(N.NikPoseResult * B)() for _ in range(3)]
    t0 = time.perf_counter(); bad = 0
    for k in range(steps):
        cf.track_batch_dev(dc.data_ptr(), list(range(B)), list(range(B, 2 * B)), True, sync=False, res=ring[k % 3])
        if k % 50 == 49:
            cf.synchronize(); bad += sum(g.as_dict() != r for g, r in zip(ring[k % 3], ref))
    cf.synchronize(); dt = time.perf_counter() - t0
    print("%dx%d polar %dx%d (family mask %d): %d steps of %d pairs, %.1f k pairs/s, mismatching results %d" % (W, H, PD, PC, cf._L.nik_is_generic(cf._ctx), steps, B, steps * B / dt / 1e3, bad), flush=True)
    cf.close()
# host frames through the tracker, many windows
H, W = 480, 640
cv = synth.canvas(7, H, W)
frames = np.stack([synth.window(cv, H, W, int(3 * i) % 200 - 100, int(2 * i) % 160 - 80, 0.5 * (i % 9)) for i in range(64)])
seq = np.stack([frames[i % 64] for i in range(4096)])
pin = torch.from_numpy(seq).pin_memory()
outs = {}
for name, kw in (("pageable", {}), ("pinned", {"ptr": pin.data_ptr()})):
    flow = N.CorrelationFlow(N.default_config(), H, W, max_batch=64, max_frames=4096 + 70)
    flow.set_kzz_cache(True)
    trk = N.Tracker(flow, N.tracker_config())
    t0 = time.perf_counter()
    o = []
    for b in range(0, 4096, 1024):                      # four calls of 16 windows each
        o += trk.push_host(seq[b:b + 1024], ptr=(kw["ptr"] + b * H * W) if kw else None)
    dt = time.perf_counter() - t0
    outs[name] = [{k: v for k, v in d.items() if k != "slot"} for d in o]
    print("push_host %s: 4096 frames %.1f k frames/s, key frames %d" % (name, 4096 / dt / 1e3, sum(d["inserted"] for d in o)), flush=True)
    trk.close(); flow.close()
print("pinned == pageable:", outs["pinned"] == outs["pageable"], " free now %.2f GB" % (torch.cuda.mem_get_info()[0] / 1e9))
