#!/usr/bin/env python
"""bench.py -- frame-pairs/s of the KCC front end (ComputeIntermedium(current) + ComputePose(key, current,
not_large_rotation=true)) on 640x480 synthetic ground texture, inputs resident in HBM.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
A step = one pass of the hot path over a batch of B frame pairs per GPU (weak scaling: B per rank fixed).
Multi-GPU: one process per GPU (torch.distributed, RCCL); pairs shard across ranks with no data-path
collective; one 4-double all-reduce per step carries the residual/PSR statistics (north_star).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

H, W, PD, PC = 480, 640, 720, 480
# algorithmic HBM bytes of one 640x480 frame pair (every 2-D FFT = 1 read + 1 write of its
# planes, all pointwise work fused, Kzz NOT cached -- the reference recomputes it per pair).
BYTES_PER_PAIR = 40.63e6
HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frame pairs per GPU per step")
    ap.add_argument("--unique", type=int, default=32, help="distinct synthetic pairs generated (tiled to --batch)")
    ap.add_argument("--cpu-sample", type=int, default=256, help="pairs timed on the host for cpu_baseline (0 = skip); 256 pairs ~ 25 core-seconds")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event pass")
    ap.add_argument("--no-cached", action="store_true", help="skip the extra Kzz-cached pass (clean rocprof traces)")
    ap.add_argument("--sequence", type=int, default=0, help="also report the tracker on a synthetic sequence of this many "
                    "frames (extra key `sequence`; not the headline metric)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import synth
    from kcc_helpers import PKG, check_pose_parity, load_module, nik
    N = nik()
    kd = load_module("kcc_dist", os.path.join(PKG, "kcc_dist.py"))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (a 1-GPU box can exercise the multi-rank path): every rank on one device, gloo instead of RCCL
    if os.environ.get("NIK_BENCH_DEVICE"):
        local_rank = int(os.environ["NIK_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("NIK_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    B, U = args.batch, min(args.unique, args.batch)
    keys_u8, curs_u8, motions = synth.make_batch(U, H, W, seed0=1000 * rank, max_shift=48, max_theta=10.0)
    reps = (B + U - 1) // U
    keys_b = np.tile(keys_u8, (reps, 1, 1))[:B]
    curs_b = np.tile(curs_u8, (reps, 1, 1))[:B]
    d_keys = torch.from_numpy(keys_b).to(dev)
    d_curs = torch.from_numpy(curs_b).to(dev)
    torch.cuda.synchronize()

    cfg = N.default_config()
    cf = N.CorrelationFlow(cfg, H, W, max_batch=B, max_frames=2 * B, device=local_rank)
    key_slots = list(range(B))
    cur_slots = list(range(B, 2 * B))
    cf.intermedium_batch_dev(d_keys.data_ptr(), B, key_slots)       # keyframe spectra: prepared before the timed region
    cf.synchronize()
    stats = torch.zeros(4, dtype=torch.float64, device=dev)

    # The library keeps two calls in flight per stream; results of step k are final once step k+2 has been queued
    # (or after synchronize()).  The per-step residual all-reduce therefore carries the statistics of step k-2.
    ring = [(N.NikPoseResult * B)() for _ in range(3)]
    state = {"k": 0}

    def step():
        k = state["k"]
        res = cf.track_batch_dev(d_curs.data_ptr(), key_slots, cur_slots, True, sync=False, res=ring[k % 3])
        if world > 1 and k >= 2:
            stats.copy_(kd.residual_stats(ring[(k - 2) % 3]), non_blocking=True)
            kd.allreduce_residual_stats(stats)                       # RCCL: [sum PSR_t, sum PSR_r, sum |t|^2, count]
        state["k"] = k + 1
        return res

    for _ in range(args.warmup):
        res = step()
    cf.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    cf.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = 1e3 * dt / args.steps
    pairs_per_s = B * world / (dt / args.steps)

    # extra (not the headline): the same workload with the per-keyframe Kzz cache (SURVEY 8d "Kzz cached", 30.17 MB/pair)
    pairs_per_s_cached = None
    if not args.no_cached:
        cf.set_kzz_cache(True)
        for _ in range(2):
            step()
        cf.synchronize()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        cf.synchronize()
        torch.cuda.synchronize()
        pairs_per_s_cached = B / ((time.perf_counter() - t1) / args.steps)       # this rank only
        cf.set_kzz_cache(False)

    out = None
    if rank == 0:
        # ---- per-kernel roofline: HIP events around every launch, on the launch stream, over extra timed steps
        roof = None
        kernels = []
        if not args.no_profile:
            nstreams = cf.set_streams(1)         # per-kernel durations are only meaningful without co-running kernels
            cf.profile_enable(True)
            psteps = max(2, min(args.steps, 5))
            for _ in range(psteps):
                cf.track_batch_dev(d_curs.data_ptr(), key_slots, cur_slots, True, sync=True)
            st = cf.profile_read()
            cf.profile_enable(False)
            cf.set_streams(int(os.environ.get("NIK_STREAMS", "2")))
            tot = sum(s["ms"] for s in st)
            for s in sorted(st, key=lambda s: -s["ms"]):
                avg_ms = s["ms"] / s["launches"]
                bpl = s["bytes"] / s["launches"]
                kernels.append(dict(name=s["name"], avg_ms=round(avg_ms, 4), share=round(s["ms"] / tot, 4),
                                    bytes_per_launch=bpl, gbps=round(bpl / (avg_ms * 1e-3) / 1e9, 1)))
            top = kernels[0]
            ach = top["bytes_per_launch"] / (top["avg_ms"] * 1e-3) / 1e9
            # measured HBM bytes of that kernel per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, corrected as
            # DESIGN.md section 4 describes; committed summary) -- only valid for the batch size it was collected at
            traffic = None
            try:
                import glob
                pm = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]))
                if pm.get("pairs_per_launch") == B:
                    traffic = pm["traffic_bytes_per_launch"].get(top["name"])
            except Exception:
                traffic = None
            roof = dict(bound="hbm", kernel=top["name"], achieved=round(ach, 1), peak=HBM_PEAK / 1e9, unit="GB/s",
                        frac=round(ach / (HBM_PEAK / 1e9), 4), traffic=traffic, avg_ms=top["avg_ms"],
                        bytes_per_launch=top["bytes_per_launch"], share_of_gpu_time=top["share"])
        # ---- CPU baseline: the oracle (a dependency-free port; the reference itself is unbuildable here).  This leg is
        # the only place bench.py touches oracle/; its per-pair outputs double as a parity spot check of the last step.
        cpu = None
        parity_ok = None
        if args.cpu_sample > 0:
            from oracle import kcc_oracle as ko
            ocfg = ko.default_config()
            ncores = os.cpu_count() or 1
            ns = args.cpu_sample
            reps_c = (ns + U - 1) // U
            kk, cc = np.tile(keys_u8, (reps_c, 1, 1))[:ns], np.tile(curs_u8, (reps_c, 1, 1))[:ns]
            # the oracle allocates plane-sized temporaries per call (like the reference's Eigen temporaries); on the
            # 2x64-core host its throughput peaks near 32 threads (tools/cpu_scale.py), so that is what is reported
            nthr = min(ncores, ns, 32)
            poses, infos, dbgs, secs_all = ko.track_pairs(ocfg, kk, cc, True, faithful=False, nthreads=nthr)
            ncheck = min(ns, B)                                          # sample pair i == pair i of the batch
            parity_ok = all(check_pose_parity(res[i].as_dict(), poses[i], infos[i], dbgs[i], PD)[0] for i in range(ncheck))
            n1 = max(1, min(8, ns))
            _, _, _, secs_1 = ko.track_pairs(ocfg, kk[:n1], cc[:n1], True, faithful=False, nthreads=1)
            _, _, _, secs_1f = ko.track_pairs(ocfg, kk[:n1], cc[:n1], True, faithful=True, nthreads=1)
            cpu = dict(value=round(ns / secs_all, 2), unit="frame-pairs/s", cores=nthr, kind="port",
                       sample="%d pairs of the same 640x480 workload, lean mode, OpenMP over pairs" % ns,
                       value_1thread=round(n1 / secs_1, 3), value_1thread_reference_faithful=round(n1 / secs_1f, 3),
                       host_cpus=ncores, gpu_results_match=bool(parity_ok), pairs_compared=ncheck)
        out = {
            "metric": "frame-pairs/s (corr-volume + pose solve) at 640x480", "value": round(pairs_per_s, 1),
            "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: 640x480 mono, ComputeIntermedium(cur)+ComputePose(key,cur,small-rot), "
                                   "polynomial kernel, polar 720x480, Kzz not cached",
                       "pairs_per_gpu_per_step": B, "unique_pairs": U, "parallelism": "pairs sharded x%d" % world,
                       "streams_per_gpu": int(os.environ.get("NIK_STREAMS", "2"))},
            "path_roofline": {"bytes_per_pair": BYTES_PER_PAIR, "achieved_GBps": round(pairs_per_s / world * BYTES_PER_PAIR / 1e9, 1),
                              "frac_of_8TBps": round(pairs_per_s / world * BYTES_PER_PAIR / HBM_PEAK, 4)},
            "roofline": roof, "cpu_baseline": cpu, "parity_spot_check": parity_ok,
            "kzz_cached_mode": None if pairs_per_s_cached is None else {
                "value_per_gpu": round(pairs_per_s_cached, 1), "bytes_per_pair": 30.17e6,
                "frac_of_8TBps": round(pairs_per_s_cached * 30.17e6 / HBM_PEAK, 4),
                "note": "same workload with the per-keyframe Kzz cache on (identical outputs); not the headline"},
            "kernels": kernels,
        }
        if args.sequence > 0:
            # configs[1] as a real sequence: the C++ tracker (MapBuilder tracking subset) with speculative batches;
            # frames between keyframe switches are registered once, the tail after a switch is re-registered.
            cv = synth.canvas(4242, H, W)
            base = [synth.window(cv, H, W, int(3 * i) % 200 - 100, int(2 * i) % 160 - 80, 0.5 * (i % 9)) for i in range(64)]
            seq = np.stack([base[i % 64] for i in range(args.sequence)])
            d_seq = torch.from_numpy(seq).to(dev)
            win = min(B, 64)
            flow2 = N.CorrelationFlow(cfg, H, W, max_batch=win, max_frames=args.sequence + win + 2, device=local_rank)
            flow2.set_kzz_cache(True)
            trk = N.Tracker(flow2, N.tracker_config())
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            outs = []
            for b0 in range(0, args.sequence, win):
                m = min(win, args.sequence - b0)
                outs += trk.push_dev(d_seq[b0:b0 + m].data_ptr(), m)
            dt_seq = time.perf_counter() - t1
            out["sequence"] = {"frames": args.sequence, "window": win, "frames_per_s": round(args.sequence / dt_seq, 1),
                               "keyframes": int(sum(o["inserted"] for o in outs)),
                               "good_tracking": int(sum(o["good_tracking"] for o in outs))}
            trk.close(); flow2.close()
        print(json.dumps(out))
    cf.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
