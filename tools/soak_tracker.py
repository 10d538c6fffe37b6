"""Soak of the tracker's look-ahead batches on the GPU (tool; tests/cpp/tracker_sched_test.cpp is the CPU counterpart against a stubbed
ABI): long sequences with IRREGULAR keyframe gaps -- many failed guesses, so batches planned on wrong guesses keep running while
their slots are recycled and rewritten -- pushed in ragged windows with and without prefetching, at look-ahead depths 1..4, small
batch rooms, with the hipGraph replay on, must give exactly the outputs of one-frame-at-a-time pushes.
usage: python tools/soak_tracker.py [frames_small [frames_full]]"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
import torch
import synth
from kcc_helpers import nik

N = nik()
KEYS = ("frame_id", "inserted", "good_tracking", "key_frame_id", "response", "cf_pose", "robot_pose", "distance")


def sequence(H, W, n, seed):
    """a random walk with bursts: speed and turn rate change every few dozen frames, with jumps"""
    rng = np.random.default_rng(seed)
    cv = synth.canvas(100 + seed, H, W)
    x = y = th = 0.0
    vx, vy, vt = 1.0, 0.5, 0.1
    out = []
    lim_x, lim_y = W // 4, H // 4
    for i in range(n):
        if i % 37 == 0 or rng.random() < 0.05:
            vx, vy, vt = rng.uniform(-3, 3) * W / 640, rng.uniform(-2, 2) * H / 480, rng.uniform(-0.6, 0.6)
        if rng.random() < 0.01:
            x, y = rng.uniform(-lim_x, lim_x), rng.uniform(-lim_y, lim_y)
        x += vx; y += vy; th += vt
        if abs(x) > lim_x: vx = -vx; x = np.clip(x, -lim_x, lim_x)
        if abs(y) > lim_y: vy = -vy; y = np.clip(y, -lim_y, lim_y)
        if abs(th) > 12: vt = -vt
        out.append(synth.window(cv, H, W, int(round(y)), int(round(x)), float(np.round(th * 2) / 2)))
    return np.stack(out)


def run(d, n, H, W, cfg, tc, window, prefetch, ragged_seed=0, graphs=0):
    flow = N.CorrelationFlow(cfg, H, W, max_batch=window, max_frames=n + 3 * window + 2)
    flow.set_kzz_cache(True); flow.set_graphs(graphs)
    trk = N.Tracker(flow, tc)
    raw = (N.NikTrackOutput * n)()
    rng = np.random.default_rng(ragged_seed)
    starts = [0]
    while starts[-1] < n:
        starts.append(min(n, starts[-1] + (int(rng.integers(1, window + 1)) if ragged_seed else window)))
    fb, base = H * W, d.data_ptr()
    t0 = time.perf_counter()
    for k in range(len(starts) - 1):
        if prefetch and k + 2 < len(starts):
            trk.prefetch_dev(base + starts[k + 1] * fb, starts[k + 2] - starts[k + 1])
        trk.push_dev_into(base + starts[k] * fb, starts[k + 1] - starts[k], raw, starts[k])
    dt = time.perf_counter() - t0
    spec = trk.speculation()
    outs = [{k: v for k, v in o.as_dict().items() if k in KEYS} for o in raw]
    trk.close(); flow.close()
    return outs, dt, spec


def soak(H, W, PD, PC, n, seeds, tc_kw):
    cfg = N.default_config(rotation_divisor=PD, rotation_channel=PC)
    tc = N.tracker_config(**tc_kw)
    total_bad = 0
    for seed in seeds:
        frames = sequence(H, W, n, seed)
        d = torch.from_numpy(frames).cuda(); torch.cuda.synchronize()
        ref, dt1, _ = run(d, n, H, W, cfg, tc, 1, False)
        nkey = sum(o["inserted"] for o in ref)
        for depth in (1, 2, 3, 4):
            for room in (0, 8):
                for (window, prefetch, rag, graphs) in ((64, True, 0, 0), (64, True, 5, 64), (32, False, 9, 0), (16, True, 3, 0)):
                    os.environ["NIK_TRK_DEPTH"] = str(depth); os.environ["NIK_TRK_FLIGHT"] = str(room)
                    got, dt, spec = run(d, n, H, W, cfg, tc, window, prefetch, rag, graphs)
                    bad = sum(a != b for a, b in zip(got, ref))
                    total_bad += bad
                    print(json.dumps(dict(geometry="%dx%d" % (W, H), seed=seed, frames=n, keyframes=nkey, depth=depth, room=room, window=window, prefetch=prefetch,
                                          ragged=bool(rag), graphs=graphs, frames_per_s=round(n / dt), guesses_held=spec[0], guesses_failed=spec[1], batches=spec[2],
                                          differing_frames=bad)), flush=True)
    return total_bad


if __name__ == "__main__":
    n_small = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    n_full = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    bad = soak(60, 80, 120, 80, n_small, (1, 2), dict(fx=75.0, fy=75.0, cx=80 / 2 - 3.5, cy=60 / 2 + 2.25, height=0.1, max_distance=0.1, max_angle=0.02,
                                                     lower_response_thr=8.0, upper_response_thr=9.0))
    bad += soak(480, 640, 720, 480, n_full, (3,), {})
    print("SOAK %s: %d differing frames" % ("FAILED" if bad else "OK", bad))
    sys.exit(1 if bad else 0)
