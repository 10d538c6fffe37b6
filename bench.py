#!/usr/bin/env python
"""bench.py -- throughput of the KCC front end on synthetic ground texture, inputs resident in HBM.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.  The default workload
("pairs", BASELINE configs[1]) is the headline: a step = ComputeIntermedium(current) + ComputePose(key, current,
not_large_rotation=true) over a batch of B 640x480 frame pairs per GPU (weak scaling: B per rank fixed).
Multi-GPU: one process per GPU; pairs shard across ranks with no data-path collective; per step the residual statistics
[sum PSR_t, sum PSR_r, sum |t|^2, count] are reduced on each device and all-reduced with RCCL through the library's own
C ABI (nik_group_allreduce_residual), asynchronously.  The timed loop is the same code for N = 1 and N > 1.

Other workloads (`--workload`, N = 1; same JSON schema, each with its own algorithmic bytes per unit from SURVEY 8(d)):
  sequence  configs[1] as a real sequence through the C++ tracker (frames/s)
  pyramid   configs[2]: 4-level coarse-to-fine, radius-4 lookup, batch 32 (pairs/s)
  hd        configs[3] geometry on one GPU: 1280x720 RGB -> luma -> pair (pairs/s)
  loop4096  configs[4]: 1 query x 4096 resident key frames, exact two-hypothesis search and top-16 short list (candidates/s)
"""
import argparse
import glob
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: 8 TB/s spec


def _streams(args, batch=0, pairs=False):
    """compute streams per GPU of this run: --streams, else $NIK_STREAMS, else 2 for big pair batches, else the library's default 3"""
    if getattr(args, "streams", 0) > 0:
        return args.streams
    if os.environ.get("NIK_STREAMS"):
        return int(os.environ["NIK_STREAMS"])
    return 2 if (pairs and batch >= 384) else 3


def _planes(H, W, PD, PC):
    N = H * W
    return N, 4.0 * N, 8.0 * (H // 2 + 1) * W, 4.0 * PD * PC, 8.0 * (PD // 2 + 1) * PC      # N, R, C, Rp, Cp


def intermedium_bytes(H, W, PD, PC):
    N, R, C, Rp, Cp = _planes(H, W, PD, PC)
    return (N + C) + (C + R) + (R + Rp) + (Rp + Cp)


def algorithmic_bytes(H, W, PD, PC, kzz_cached=False, hypotheses=1, with_intermedium=True):
    """SURVEY.md 8(d) accounting: every 2-D FFT = one read of its input plane(s) + one write of its output plane, all
    pointwise work fused, gathers read their source once; Kzz recomputed per pair unless kzz_cached."""
    N, R, C, Rp, Cp = _planes(H, W, PD, PC)

    def stage(r, c):                       # EstimateTrans: Kzz (2 FFTs), Kxz (2 FFTs, two input spectra), g (1 FFT)
        return (0.0 if kzz_cached else 2 * (r + c)) + (2 * (r + c) + c) + (r + c)
    return (intermedium_bytes(H, W, PD, PC) if with_intermedium else 0.0) + stage(Rp, Cp) + hypotheses * ((R + C) + stage(R, C))


def _profile(cf, run_once, steps, restore_streams=None):
    """per-kernel HIP-event timings on ONE stream (durations are only meaningful without co-running kernels)"""
    cf.set_streams(1)
    cf.profile_enable(True)
    for _ in range(steps):
        run_once()
    cf.synchronize()
    st = cf.profile_read()
    cf.profile_enable(False)
    cf.set_streams(restore_streams or int(os.environ.get("NIK_STREAMS", "3")))
    tot = sum(s["ms"] for s in st) or 1.0
    kernels = []
    for s in sorted(st, key=lambda s: -s["ms"]):
        if not s["launches"]:
            continue
        avg_ms = s["ms"] / s["launches"]
        bpl = s["bytes"] / s["launches"]
        bdl = s.get("bytes_design", s["bytes"]) / s["launches"]
        # bytes_per_launch: the pass's nominal planes (SURVEY 8d accounting); design_bytes_per_launch: what the launch is built
        # to move (symmetry shortcuts taken out).  gbps is priced on the DESIGN bytes: a kernel must never be credited with bytes
        # it does not touch (round 5 listed kA_inv<.,shifted> at 7.9-10.9 TB/s on nominal planes it only reads 40 % of).
        k = dict(name=s["name"], avg_ms=round(avg_ms, 4), share=round(s["ms"] / tot, 4), bytes_per_launch=bpl,
                 design_bytes_per_launch=bdl, gbps=round(bdl / (avg_ms * 1e-3) / 1e9, 1), gbps_nominal=round(bpl / (avg_ms * 1e-3) / 1e9, 1))
        if k["gbps"] > HBM_PEAK / 1e9 and not os.environ.get("NIK_ABLATE"):      # (an ablated kernel moves nothing: tools/ablate.sh)
            raise AssertionError("%s: %.1f GB/s on its design bytes exceeds the %.0f GB/s HBM peak -- the byte accounting of this stage is wrong"
                                 % (k["name"], k["gbps"], HBM_PEAK / 1e9))
        kernels.append(k)
    return kernels


def _live_rocprof(args, workload, batch):
    """rocprofv3 passes over a short run of this very bench (one stream, no nested profiling): per-kernel durations from
    --kernel-trace --stats and HBM bytes from the FETCH_SIZE / WRITE_SIZE passes.  Any failure returns None (the line then
    falls back on the committed profiles/ summary and says so)."""
    if args.no_live_prof or shutil.which("rocprofv3") is None:
        return None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import rocprof_summary
        out = os.path.join(ROOT, "gpurun_out", "bench_live_prof")
        # the passes run THIS bench with its HIP-event pass switched on: the rocprofv3 durations and HIP-event durations of ONE
        # process can then be compared (the method check), next to the main process's own (the HBM-bound kernels sit on
        # different levels from process to process: profiles/r06_placement_probe.txt)
        dump = os.path.join(out, "hip_events_in_pass")
        for f in glob.glob(dump + ".*"):
            os.remove(f)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--batch", str(batch), "--steps", "120", "--warmup", "40",
               "--cpu-sample", "0", "--no-cached", "--no-live-prof", "--repeats", "1", "--kernels-dump", dump]
        r = rocprof_summary.collect(cmd, out, want_pmc=True, timeout=240)
        dumps = sorted(glob.glob(dump + ".*"), key=os.path.getmtime)
        if dumps:                                             # (the first pass is the --kernel-trace --stats one)
            r["hip_events_same_process"] = {k["name"]: k["avg_ms"] for k in json.load(open(dumps[0]))}
        return r if r["stats"] else None
    except Exception as e:                                    # noqa: BLE001
        sys.stderr.write("live rocprof pass failed: %s\n" % str(e)[:300])
        return None


def _roofline(kernels, batch, live=None, contract_bytes_per_unit=None):
    """The kernel with the largest share of GPU time against the HBM roofline.
    achieved = ALGORITHMIC bytes of a launch / its average duration.  The duration is measured twice: HIP events
    on the launch stream inside this process (avg_ms_hip_event) and rocprofv3 --kernel-trace --stats (avg_ms_rocprof: a live
    pass of this run when rocprofv3 is present, else the committed profiles/<tag>_kernel_times.json).  frac is priced on the
    LARGER of the two, and the line says whether they agree within 5 %.
    traffic = the kernel's measured HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (live, else the committed
    summary -- traffic_source says which); frac_moved_bytes prices the kernel on those bytes."""
    if not kernels:
        return None
    top = kernels[0]
    name = top["name"]
    t_hip = top["avg_ms"]
    t_roc, roc_src, traffic, tr_src, t_hip_same = None, None, None, None, None
    if live is not None:
        t_hip_same = (live.get("hip_events_same_process") or {}).get(name)
        if name in live["stats"]:
            t_roc, roc_src = live["stats"][name]["avg_ms"], "live rocprofv3 --kernel-trace --stats pass of this run"
        if name in live["traffic"]:
            traffic, tr_src = live["traffic"][name], "live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run"
    try:
        if t_roc is None:
            f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_times.json")))[-1]
            kt = json.load(open(f))
            if kt.get("pairs_per_launch") == batch and name in kt["kernels"]:
                t_roc, roc_src = kt["kernels"][name]["avg_ms"], "profiles/" + os.path.basename(f)
        if traffic is None:
            f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
            pm = json.load(open(f))
            if pm.get("pairs_per_launch") == batch:
                traffic, tr_src = pm["traffic_bytes_per_launch"].get(name), "profiles/" + os.path.basename(f) + " (committed, not re-measured)"
    except Exception:                                         # noqa: BLE001
        pass
    t_use = max([t for t in (t_hip, t_roc, t_hip_same) if t])
    ach = top["bytes_per_launch"] / (t_use * 1e-3) / 1e9
    r = dict(bound="hbm", kernel=name, achieved=round(ach, 1), peak=HBM_PEAK / 1e9, unit="GB/s", frac=round(ach / (HBM_PEAK / 1e9), 4),
             traffic=traffic, traffic_source=tr_src, avg_ms=round(t_use, 4), avg_ms_hip_event=t_hip,
             avg_ms_rocprof=None if t_roc is None else round(t_roc, 4), rocprof_source=roc_src,
             bytes_per_launch=top["bytes_per_launch"], share_of_gpu_time=top["share"])
    if t_roc:
        # Two comparisons.  (1) the METHOD: HIP events against rocprofv3 inside ONE process (the rocprofv3 pass runs this bench with
        # its HIP-event pass on) -- that is what durations_agree_within_5pct reports when the pass delivered it.  (2) this process
        # against that one: the HBM-bound kernels sit on discrete levels that change from process to process (up to +- 8 %,
        # profiles/r06_placement_probe.txt), so this ratio is reported, not asserted.  frac is priced on the LARGEST duration seen.
        r["hip_event_over_rocprof"] = round(t_hip / t_roc, 4)
        if t_hip_same:
            r["avg_ms_hip_event_in_rocprof_process"] = round(t_hip_same, 4)
            r["hip_event_over_rocprof_same_process"] = round(t_hip_same / t_roc, 4)
            r["durations_agree_within_5pct"] = bool(abs(t_hip_same / t_roc - 1.0) <= 0.05)
            r["durations_compared"] = "HIP events and rocprofv3 --stats of the same process (the live pass); hip_event_over_rocprof compares THIS process with that one"
        else:
            r["durations_agree_within_5pct"] = bool(abs(t_hip / t_roc - 1.0) <= 0.05)
            r["durations_compared"] = "HIP events of this process against rocprofv3 --stats of another run"
        if not r["durations_agree_within_5pct"]:
            sys.stderr.write("WARNING: %s: HIP-event %.4f ms vs rocprof %.4f ms per launch disagree by more than 5 %%; roofline.frac uses the larger\n"
                             % (name, t_hip_same or t_hip, t_roc))
    if traffic:
        r["frac_moved_bytes"] = round(traffic / (t_use * 1e-3) / HBM_PEAK, 4)
    # the same kernel on the bytes it is built to move (Hermitian-half / trimmed planes taken out of the nominal count)
    r["design_bytes_per_launch"] = top.get("design_bytes_per_launch", top["bytes_per_launch"])
    r["frac_design"] = round(r["design_bytes_per_launch"] / (t_use * 1e-3) / HBM_PEAK, 4)
    if contract_bytes_per_unit:
        # The per-kernel nominal bytes of this two-pass design sum to MORE than the contract's bytes per unit (SURVEY 8d counts a
        # 2-D FFT as one read + one write).  frac_contract deflates the dominant kernel's fraction by that generosity, so that it
        # is comparable with path_roofline.
        gen = sum(k["bytes_per_launch"] for k in kernels) / batch / contract_bytes_per_unit
        r["nominal_bytes_over_contract"] = round(gen, 4)
        r["frac_contract"] = round(r["frac"] / gen, 4)
    return r


def _line(metric, unit, value, world, args, ms_per_step, workload, bytes_per_unit, extra_cfg=None, streams=None, **more):
    out = {"metric": metric, "value": round(value, 1), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": dict({"workload": workload, "streams_per_gpu": streams or int(os.environ.get("NIK_STREAMS", "3"))}, **(extra_cfg or {})),
           "path_roofline": {"bytes_per_unit": bytes_per_unit, "achieved_GBps": round(value / world * bytes_per_unit / 1e9, 1),
                             "frac_of_8TBps": round(value / world * bytes_per_unit / HBM_PEAK, 4)}}
    out.update(more)
    return out


def _make_group(N, torch, dist, np, cf, rank, world, dev, what):
    """the library's nik_group for this rank (RCCL inside the C ABI).  Returns (group or None, description, fallback, stats_t):
    with the test hook NIK_BENCH_BACKEND=gloo (several ranks on one device, where RCCL cannot form a communicator) or when the
    group cannot be created, the exchange runs through torch.distributed instead and the line says so."""
    comm, grp, stats_t, fallback = "nik_group (single GPU: no collective)", None, None, False
    if world > 1 and os.environ.get("NIK_BENCH_BACKEND", "nccl") != "nccl":
        cf.set_residual_stats(True)
        stats_t = torch.zeros(4, dtype=torch.float64)
        comm = "device-side reduction + torch.distributed(%s) %s [test hook]" % (os.environ["NIK_BENCH_BACKEND"], what)
        return grp, comm, fallback, stats_t
    try:
        uid = None
        if world > 1:
            t = torch.from_numpy(N.Group.unique_id() if rank == 0 else np.zeros(128, np.uint8)).to(dev)
            dist.broadcast(t, src=0)
            uid = t.cpu().numpy()
            comm = "nik_group: RCCL %s inside the C ABI" % what
        grp = N.Group.rank(cf, rank, world, uid)
    except Exception as e:                                   # keep the measurement alive; say so in the line
        grp, fallback = None, True
        cf.set_residual_stats(True)
        stats_t = torch.zeros(4, dtype=torch.float64, device=dev)
        comm = "FALLBACK torch.distributed %s (nik_group failed: %s)" % (what, str(e)[:200])
    if world > 1:
        # every rank must take the same road: a communicator that formed on some ranks only would leave them waiting in RCCL
        # for the ranks that fell back to torch.distributed
        ok = torch.tensor([0 if fallback else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and grp is not None:
            grp.close()
            grp, fallback = None, True
            cf.set_residual_stats(True)
            stats_t = torch.zeros(4, dtype=torch.float64, device=dev)
            comm = "FALLBACK torch.distributed %s (nik_group failed on another rank)" % what
    return grp, comm, fallback, stats_t


def _multi_gpu_facts(N, world, grp, fallback, **more):
    lib, shared = N.Group.rccl_library() if (grp is not None and world > 1) or os.environ.get("NIK_GROUP_FORCE_RCCL") else (None, False)
    return dict(dict(world=world, rccl_ranks=grp.comm_ranks() if grp is not None else 0, fallback=bool(fallback),
                     rccl_library=lib, rccl_shared_with_torch=bool(shared)), **more)


# ---------------------------------------------------------------------------------------------------------------------
def workload_pairs(args, N, torch, dist, np, synth, world, rank, dev, local_rank, hd=False):
    """The headline workload (hd=False: 640x480 gray pairs) and configs[3] (hd=True: 1280x720 RGB frames -> integer luma ->
    the same pair unit) share this code: one process per GPU, the pairs of a step sharded over the ranks, the residual
    statistics reduced on every device and all-reduced through nik_group (RCCL inside the C ABI)."""
    from kcc_helpers import imposed_rerun, parity_detail, parity_summary
    H, W, PD, PC = (720, 1280, 720, 480) if hd else (480, 640, 720, 480)
    B = args.batch
    if hd and not args.batch_given:
        B = 128 if world == 1 else 32            # configs[3]: "batch 256 frame-pairs sharded across 8 GPUs" = 32 per GPU
    U = B if args.unique <= 0 else min(args.unique, B)
    if hd and args.unique <= 0:
        U = min(B, 32)                           # (a 1280x720 canvas takes seconds to synthesise)
    gb = int(os.environ.get("NIK_BENCH_GLOBAL_BATCH", "0"))
    mk = dict(max_theta=8.0, max_shift=60, ncanvas=8) if hd else dict(max_theta=10.0)
    if gb:      # test hook: ONE global batch, this rank takes its contiguous shard (sharded == unsharded can then be checked)
        gk, gc_, _ = synth.make_unique_batch(gb, H, W, seed0=777, **mk)
        b0, e0 = N.Group.shard(gb, world, rank)
        keys_u8, curs_u8, B = gk[b0:e0], gc_[b0:e0], e0 - b0
        U = B
    else:
        keys_u8, curs_u8, motions = synth.make_unique_batch(U, H, W, seed0=1000 * (rank + 1), **mk)
    reps = (B + U - 1) // U
    cfg = N.default_config()
    cf = N.CorrelationFlow(cfg, H, W, max_batch=B, max_frames=2 * B, device=local_rank)
    n_streams = cf.set_streams(_streams(args, B, pairs=not hd))
    if args.chunk > 0:
        cf.set_chunk(args.chunk)
    key_slots, cur_slots = list(range(B)), list(range(B, 2 * B))
    d_rgb = None
    if hd:
        # frames arrive as RGB (R = G = B = the synthetic texture, so the integer luma returns it exactly and the oracle can
        # be fed the gray images); the colour conversion of the current frames is part of every step
        to_rgb = lambda a: torch.from_numpy(np.repeat(np.tile(a, (reps, 1, 1))[:B, :, :, None], 3, axis=3).copy()).to(dev)   # noqa: E731
        d_keys_rgb, d_rgb = to_rgb(keys_u8), to_rgb(curs_u8)
        d_keys = torch.empty((B, H, W), dtype=torch.uint8, device=dev); d_curs = torch.empty_like(d_keys)
        torch.cuda.synchronize()
        cf.rgb_to_gray_dev(d_keys_rgb.data_ptr(), B, d_keys.data_ptr())
        del d_keys_rgb
    else:
        d_keys = torch.from_numpy(np.tile(keys_u8, (reps, 1, 1))[:B]).to(dev)
        d_curs = torch.from_numpy(np.tile(curs_u8, (reps, 1, 1))[:B]).to(dev)
    torch.cuda.synchronize()
    cf.intermedium_batch_dev(d_keys.data_ptr(), B, key_slots)       # keyframe spectra: prepared before the timed region
    cf.synchronize()

    # the residual all-reduce: through the library's nik_group (RCCL inside the C ABI).  Test hook NIK_BENCH_BACKEND=gloo
    # (several ranks on one device, where RCCL cannot form a communicator): device-side reduction + torch.distributed.
    grp, comm, fallback, stats_t = _make_group(N, torch, dist, np, cf, rank, world, dev, "all-reduce of 4 doubles per step")

    # The library keeps two calls in flight per stream; results of step k are final once step k+2 has been queued (or after
    # synchronize()).  Nothing in the loop touches per-pair results on the host.
    ring = [(N.NikPoseResult * B)() for _ in range(3)]
    state = {"k": 0}

    rccl_ranks = grp.comm_ranks() if grp is not None else 0       # ncclCommCount of the library's communicator (0: none in use)

    def step():
        k = state["k"]
        if d_rgb is not None:
            cf.rgb_to_gray_async(d_rgb.data_ptr(), B, d_curs.data_ptr())
        res = cf.track_batch_dev(d_curs.data_ptr(), key_slots, cur_slots, True, sync=False, res=ring[k % 3])
        if grp is not None:
            grp.allreduce_residual(wait=False)                   # device-side reduction + RCCL, asynchronous
        elif world > 1:
            stats_t.copy_(torch.from_numpy(cf.residual_stats()))
            dist.all_reduce(stats_t)
        state["k"] = k + 1
        return res

    for _ in range(args.warmup):
        res = step()
    cf.synchronize()

    def timed_region():
        """EXACTLY args.steps steps between barrier + synchronize on both sides; returns this rank's seconds"""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = None
        for _ in range(args.steps):
            r = step()
        cf.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return time.perf_counter() - t0, r

    # The timed region (K steps, ~50 ms at the defaults) is repeated -- the same K steps every time -- until a second of
    # measurement has accumulated: one region alone cannot show a 3 % change (boxes and clock ramps differ by more).  The
    # line reports the MEDIAN region (value, ms_per_step) and the fastest / slowest one next to it; with more than one rank
    # every region's time is the MAX over the ranks.
    cdev = dev if (world > 1 and dist.get_backend() == "nccl") else "cpu"
    region_max, region_own = [], []
    n_rep = 1 if args.repeats == 1 else None
    while True:
        dt_own, res = timed_region()
        dt_max = dt_own
        if world > 1:
            t = torch.tensor([dt_own], dtype=torch.float64, device=cdev)
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            region_own.append([float(x.item()) for x in allt])
            dt_max = max(region_own[-1])
        else:
            region_own.append([dt_own])
        region_max.append(dt_max)
        if n_rep is None:                        # number of repeats: fixed by the first region (identical on every rank)
            n_rep = args.repeats if args.repeats > 0 else int(min(64, max(3, round(args.min_time / max(dt_max, 1e-6)))))
        if len(region_max) >= n_rep:
            break
    order = sorted(range(len(region_max)), key=lambda i: region_max[i])
    med = order[len(order) // 2]
    dt = region_max[med]
    stats = grp.residual_result() if grp is not None else (stats_t.cpu().numpy() if stats_t is not None else None)
    rank_rates = [B / (x / args.steps) for x in region_own[med]]                 # every rank's own pairs/s over its own clock (median region)
    ms_per_step = 1e3 * dt / args.steps
    timing = dict(regions=len(region_max), steps_per_region=args.steps, ms_per_step_median=round(ms_per_step, 4),
                  ms_per_step_min=round(1e3 * min(region_max) / args.steps, 4), ms_per_step_max=round(1e3 * max(region_max) / args.steps, 4),
                  value_max=round(B * world / (min(region_max) / args.steps), 1), value_min=round(B * world / (max(region_max) / args.steps), 1),
                  note="value / ms_per_step are the median timed region of K steps; every region is bracketed by barrier + synchronize and is the MAX over ranks")
    pairs_per_s = B * world / (dt / args.steps)
    last = [r.as_dict() for r in res]                               # the final timed step's per-pair results (this rank)

    # extra (not the headline): the same workload with the per-keyframe Kzz cache (SURVEY 8d "Kzz cached")
    pairs_per_s_cached = None
    if not args.no_cached and world == 1:
        cf.set_kzz_cache(True)
        for _ in range(2):
            step()
        cf.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        cf.synchronize()
        pairs_per_s_cached = B / ((time.perf_counter() - t1) / args.steps)
        cf.set_kzz_cache(False)

    if os.environ.get("NIK_BENCH_DUMP"):                            # test hook: this rank's results of the last timed step
        json.dump(dict(rank=rank, world=world, results=last, stats=None if stats is None else [float(v) for v in stats]),
                  open(os.environ["NIK_BENCH_DUMP"] + ".%d" % rank, "w"))
    out = None
    if rank == 0:
        # (asynchronous calls, as in the timed region, and enough of them: five synchronous steps left the clocks of a freshly
        # idle GPU in the numbers and disagreed with rocprofv3 by up to 6 %)
        kernels = [] if args.no_profile else _profile(cf, lambda: cf.track_batch_dev(d_curs.data_ptr(), key_slots, cur_slots, True, sync=False, res=ring[0]),
                                                       max(2, min(args.steps, 20)), n_streams)
        if args.kernels_dump and kernels:
            json.dump(kernels, open("%s.%d" % (args.kernels_dump, os.getpid()), "w"))
        # the live rocprofv3 passes run NOW, while the part is as warm as it was under the HIP-event pass above (behind the CPU
        # leg below it would have idled for ~15 s: the HBM-bound dominant kernel then measured up to 7 % off, in either direction)
        live = _live_rocprof(args, "hd" if hd else "pairs", B) if (world == 1 and kernels) else None
        # ---- CPU baseline: the oracle (a dependency-free port; the reference itself is unbuildable here).  This leg is the
        # only place bench.py touches oracle/; its per-pair outputs double as a parity spot check of the last timed step.
        cpu, parity_ok, parity = None, None, None
        if args.cpu_sample > 0 and world == 1:
            from oracle import kcc_oracle as ko
            ocfg = ko.default_config()
            ncores = os.cpu_count() or 1
            ns = min(args.cpu_sample, U, 16 if hd else U)
            # on the 2x64-core host the oracle's throughput peaks at 16-32 threads and FALLS beyond (tools/cpu_scale.py: 11 / 94 /
            # 165 / 197 / 159 / 137 / 92 pairs/s at 1 / 8 / 16 / 32 / 64 / 128 / 256 threads, profiles/r05_cpu_scale.txt).  Round 5
            # gave every thread one arena for its plane-sized temporaries (no malloc / mmap / page faults in the timed units): the
            # curve did not move (r04: 12 / 96 / 195 / 180 / 166 / 138 / 88), so it is the memory system -- 10-30 MB of strided planes
            # per thread against 32 MB of L3 per 8 cores -- not the allocator.  The peak is what is reported.
            nthr = min(ncores, ns, 32)
            poses, infos, dbgs, secs_all = ko.track_pairs(ocfg, keys_u8[:ns], curs_u8[:ns], True, faithful=False, nthreads=nthr)
            parity = parity_summary([parity_detail(last[i], poses[i], infos[i], dbgs[i], PD,
                                                   rerun=imposed_rerun(ocfg, H, W, keys_u8[i], curs_u8[i], True)) for i in range(ns)])
            parity_ok = parity["ok"]
            n1 = max(1, min(8, ns))
            _, _, _, secs_1 = ko.track_pairs(ocfg, keys_u8[:n1], curs_u8[:n1], True, faithful=False, nthreads=1)
            _, _, _, secs_1f = ko.track_pairs(ocfg, keys_u8[:n1], curs_u8[:n1], True, faithful=True, nthreads=1)
            cpu = dict(value=round(ns / secs_all, 2), unit="frame-pairs/s", cores=nthr, kind="port",
                       sample="%d unique pairs of the same %dx%d workload, lean mode, OpenMP over pairs" % (ns, W, H),
                       value_1thread=round(n1 / secs_1, 3), value_1thread_reference_faithful=round(n1 / secs_1f, 3),
                       host_cpus=ncores, gpu_results_match=bool(parity_ok), pairs_compared=ns)
        bpp = algorithmic_bytes(H, W, PD, PC)
        bpc = algorithmic_bytes(H, W, PD, PC, kzz_cached=True)
        metric = "frame-pairs/s at 1280x720 RGB (configs[3])" if hd else "frame-pairs/s (corr-volume + pose solve) at 640x480"
        wl = ("configs[3]: 1280x720 RGB -> integer luma -> ComputeIntermedium(cur)+ComputePose(key,cur,small-rot), %d pairs per GPU per step, pairs sharded over the GPUs, RCCL residual all-reduce"
              % B) if hd else "configs[1]: 640x480 mono, ComputeIntermedium(cur)+ComputePose(key,cur,small-rot), polynomial kernel, polar 720x480, Kzz not cached"
        out = _line(metric, "frame-pairs/s", pairs_per_s, world, args, ms_per_step, wl,
                    bpp, dict(pairs_per_gpu_per_step=B, unique_pairs=U, parallelism="pairs sharded x%d" % world, residual_allreduce=comm), streams=n_streams,
                    roofline=_roofline(kernels, B, live, bpp), cpu_baseline=cpu, parity_spot_check=parity,
                    residual_stats=None if stats is None else [float(v) for v in stats], timing=timing,
                    multi_gpu=_multi_gpu_facts(N, world, grp, fallback,
                                   pairs_per_s_per_rank_min=round(min(rank_rates), 1), pairs_per_s_per_rank_max=round(max(rank_rates), 1),
                                   note="rccl_ranks = ncclCommCount of the library's communicator (0: one GPU, no collective); fallback: the residual all-reduce ran through torch.distributed instead of nik_group; rccl_library: the file the library's ncclAllReduce came from, rccl_shared_with_torch: it is the copy PyTorch had loaded (RTLD_NOLOAD)"),
                    kzz_cached_mode=None if pairs_per_s_cached is None else {
                        "value_per_gpu": round(pairs_per_s_cached, 1), "bytes_per_pair": bpc,
                        "frac_of_8TBps": round(pairs_per_s_cached * bpc / HBM_PEAK, 4),
                        "note": "same workload with the per-keyframe Kzz cache on (identical outputs); not the headline"},
                    kernels=kernels,
                    kernels_note=None if not kernels else (
                        "bytes_per_launch = the nominal planes of each pass of this two-pass (A: lines along the halved axis, B: spectrum lines) design; they sum to %.2f MB per pair, "
                        "%.3fx the contract's %.2f MB (SURVEY 8d counts a 2-D FFT as one read + one write): path_roofline is priced on the contract's bytes, roofline.achieved on the dominant "
                        "kernel's nominal bytes, roofline.frac_contract deflates it by that factor.  design_bytes_per_launch = the bytes a launch is built to move (Hermitian-half Kzz plane, "
                        "trimmed zero-phase columns taken out): %.2f MB per pair; per-kernel gbps is priced on THOSE and asserted <= the 8000 GB/s peak (gbps_nominal: on the nominal planes)"
                        % (sum(k["bytes_per_launch"] for k in kernels) / B / 1e6, sum(k["bytes_per_launch"] for k in kernels) / B / bpp, bpp / 1e6,
                           sum(k["design_bytes_per_launch"] for k in kernels) / B / 1e6)))
        out["path_roofline"]["bytes_per_pair"] = bpp
    if grp is not None:
        grp.close()
    cf.close()
    return out


def workload_sequence(args, N, torch, np, synth, dev, local_rank):
    """configs[1] as a real sequence: the C++ tracker (MapBuilder tracking subset) with speculative batches; frames between
    keyframe switches are registered once, the tail after a switch is re-registered."""
    H, W, PD, PC = 480, 640, 720, 480
    T = args.frames
    cv = synth.canvas(4242, H, W)
    if args.seq_motion == "smooth":
        # constant speed along a zig-zag (3 px and 2 px per frame, --seq-rot-rate degrees per frame, all three turning together
        # every 64 frames): what a ground robot's down-looking camera sees; keyframes then come at a near-regular spacing
        # (the 3-degree rule: every 3 / rate frames)
        tri = lambda i, half: abs(((i + half) % (2 * half)) - half)               # 0 .. half .. 0
        base = [synth.window(cv, H, W, 3 * tri(i, 64) - 96, 2 * tri(i, 64) - 64, args.seq_rot_rate * tri(i, 64)) for i in range(128)]
    else:
        # saw-tooth motion: the rotation jumps back by 4 degrees every 9 frames and the view jumps every 64 -- keyframes at
        # irregular spacing (the tracker's guesses of the next keyframe mostly fail)
        base = [synth.window(cv, H, W, int(3 * i) % 200 - 100, int(2 * i) % 160 - 80, 0.5 * (i % 9)) for i in range(64)]
    seq = np.stack([base[i % len(base)] for i in range(T)])
    d_seq = torch.from_numpy(seq).to(dev)
    win = min(args.batch, int(os.environ.get("NIK_SEQ_WINDOW", "64")))
    cfg = N.default_config()

    def run(nframes, graphs=0, prefetch=True):
        flow = N.CorrelationFlow(cfg, H, W, max_batch=win, max_frames=nframes + win + 2, device=local_rank)
        flow.set_kzz_cache(True)
        flow.set_graphs(graphs)
        trk = N.Tracker(flow, N.tracker_config())
        torch.cuda.synchronize()
        raw = (N.NikTrackOutput * nframes)()                    # the C caller's output array (dicts are made after the clock stops)
        base, fb = d_seq.data_ptr(), H * W
        t1 = time.perf_counter()
        for b0 in range(0, nframes, win):
            m = min(win, nframes - b0)
            if prefetch and b0 + m < nframes:                   # the next window's spectra run beside this window's registrations
                trk.prefetch_dev(base + (b0 + m) * fb, min(win, nframes - b0 - m))
            trk.push_dev_into(base + b0 * fb, m, raw, b0)
        dt = time.perf_counter() - t1
        outs = [o.as_dict() for o in raw]
        spec_box[:] = trk.speculation(); stats_box.update(trk.stats())
        trk.close(); flow.close()
        return outs, dt
    def run_host(nframes, src, ptr=None):
        """the same sequence from HOST memory (nik_tracker_push_host): windows of `win` frames, the next window uploaded on
        the context's upload stream while the current one is registered"""
        flow = N.CorrelationFlow(cfg, H, W, max_batch=win, max_frames=nframes + win + 2, device=local_rank)
        flow.set_kzz_cache(True)
        trk = N.Tracker(flow, N.tracker_config())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        raw = trk.push_host(src[:nframes], ptr=ptr, raw=True)
        dt = time.perf_counter() - t1
        outs = [o.as_dict() for o in raw]
        trk.close(); flow.close()
        return outs, dt
    spec_box = [0, 0, 0]; stats_box = {}
    run(min(T, 256))                                            # warm-up (module load, first launches)
    best, best_g = None, None
    for _ in range(max(1, args.steps // 10)):
        outs, dt = run(T)
        best = dt if best is None else min(best, dt)
        outs_g, dt = run(T, graphs=win)                         # small batches replayed as hipGraphs (nik_set_graphs)
        best_g = dt if best_g is None else min(best_g, dt)
    graphs_same = all(all(a[k] == b[k] for k in a if k != "slot") for a, b in zip(outs, outs_g))
    use_g, best_off = best_g < best, best
    if use_g:
        best, outs = best_g, outs_g
    # property check (size-independent): pushing the frames one by one gives the same decisions as the speculative windows
    parity = None
    if args.cpu_sample > 0:
        ns = min(T, 2048)                          # (every frame: a wrong keyframe guess that slipped through would show here)
        flow = N.CorrelationFlow(cfg, H, W, max_batch=1, max_frames=ns + 3, device=local_rank)
        flow.set_kzz_cache(True)
        trk = N.Tracker(flow, N.tracker_config())
        one = []
        for i in range(ns):
            one += trk.push_dev(d_seq[i:i + 1].data_ptr(), 1)
        trk.close(); flow.close()
        same = lambda a, b: all(a[k] == b[k] for k in a if k != "slot")          # (slot numbers depend on the window size)
        parity = all(same(one[i], outs[i]) for i in range(ns))
    host = None
    if args.host_frames:
        # host-inclusive rate (the reference's caller hands over host images: main.cpp:55-65): (a) frames in pinned memory -- a
        # camera driver's DMA ring -- read by the copy engine directly, (b) frames in pageable memory, staged through the
        # context's pinned buffers by the calling thread
        pin = torch.from_numpy(seq).pin_memory()
        run_host(min(T, 256), seq)
        bp, bq, same_p, same_q = None, None, True, True
        for _ in range(max(1, args.steps // 10)):
            o1, dt1 = run_host(T, pin.numpy(), ptr=pin.data_ptr()); bp = dt1 if bp is None else min(bp, dt1)
            o2, dt2 = run_host(T, seq); bq = dt2 if bq is None else min(bq, dt2)
            same = lambda a, b: all(a[k] == b[k] for k in a if k != "slot")         # noqa: E731
            same_p = same_p and all(same(a, b) for a, b in zip(o1, outs)); same_q = same_q and all(same(a, b) for a, b in zip(o2, outs))
        host = {"frames_per_s_pinned_source": round(T / bp, 1), "frames_per_s_pageable_source": round(T / bq, 1),
                "frames_per_s_resident": round(T / best_off, 1), "identical_outputs": bool(same_p and same_q),
                "upload_GBps_pinned": round(T / bp * H * W / 1e9, 2),
                "note": "nik_tracker_push_host: uploads on the context's own stream (never a compute lane), window k+1 travels while window k is registered; pageable sources are copied into pinned staging by the calling thread"}
    nkey = int(sum(o["inserted"] for o in outs))
    bpf = algorithmic_bytes(H, W, PD, PC, kzz_cached=True)
    return _line("frames/s through the tracker (configs[1] as a sequence)", "frames/s", T / best, 1, args, 1e3 * best,
                 "configs[1] sequence: %d frames (%s camera path), C++ tracker (keyframe rule, PSR gating), windows of %d frames, look-ahead batches along the guessed keyframe chain, Kzz cached per keyframe" % (T, args.seq_motion, win),
                 bpf, dict(frames=T, window=win, keyframes=nkey, good_tracking=int(sum(o["good_tracking"] for o in outs)),
                           keyframe_guesses_held=spec_box[0], keyframe_guesses_failed=spec_box[1], batched_pose_calls=spec_box[2], registrations=stats_box),
                 parity_spot_check=parity, roofline=None, cpu_baseline=None,
                 hipgraph={"frames_per_s_off": round(T / best_off, 1), "frames_per_s_on": round(T / best_g, 1),
                           "identical_outputs": bool(graphs_same), "reported": "on" if use_g else "off"},
                 host_inclusive=host,
                 note="the tracker guesses the coming keyframes from the history of their gaps (nik_tracker_guess_gap) and keeps asynchronous pose batches planned along that chain in flight while it applies the previous one; a result is used only if its key is the frame the reference's rule really inserted: all outputs equal one-frame-at-a-time pushes (parity_spot_check compares every frame)")


def workload_pyramid(args, N, torch, np, synth, dev, local_rank):
    H, W, B, LEVELS, R = 480, 640, min(args.batch, 32), 4, 4
    pyr = N.Pyramid(N.default_config(), H, W, levels=LEVELS, max_batch=B, device=local_rank)
    keys, curs, _ = synth.make_unique_batch(B, H, W, seed0=50, max_theta=8.0, max_shift=40)
    dk = torch.from_numpy(keys).to(dev); dc = torch.from_numpy(curs).to(dev)
    torch.cuda.synchronize()
    # batches are enqueued back to back (the coarse levels of batch k+1 run beside the fine levels of batch k); results
    # of batch k are final once batch k+2 has been enqueued, and all of them after the synchronize inside the timed region
    ring = [(N.NikPoseResult * (LEVELS * B))() for _ in range(3)]
    for k in range(args.warmup):
        pyr.track_dev_async(dk.data_ptr(), dc.data_ptr(), B, R, res=ring[k % 3])
    pyr.synchronize()
    # (a batch takes ~1.3 ms from its first launch to its last result while a new one starts every ~0.8 ms: time at least 100 of
    # them, or the drain of the pipeline at the end is 8 % of the measurement)
    import copy
    args = copy.copy(args); args.steps = max(args.steps, 100)
    t0 = time.perf_counter()
    for k in range(args.steps):
        raw = pyr.track_dev_async(dk.data_ptr(), dc.data_ptr(), B, R, res=ring[k % 3])
    pyr.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    res = pyr.as_lists(raw, B)
    sync_res = pyr.track_dev(dk.data_ptr(), dc.data_ptr(), B, R)               # one batch alone: same answers
    pipelined_equals_single = all(res[l][i]["pose"] == sync_res[l][i]["pose"] for l in range(LEVELS) for i in range(B))
    parity = None
    if args.cpu_sample > 0:
        from oracle import kcc_oracle as ko
        ns = min(4, B)
        poses, _, _, _ = ko.track_pairs(ko.default_config(), keys[:ns], curs[:ns], True, nthreads=ns)
        parity = all(res[0][i]["pose"][0] == poses[i][0] and res[0][i]["pose"][1] == poses[i][1] for i in range(ns))   # level 0 == plain KCC
    # per pair: key AND current intermedium plus one pose at every level
    bpp = sum(intermedium_bytes(h, w, pd, pc) + algorithmic_bytes(h, w, pd, pc) for (h, w, pd, pc) in pyr.dims)
    out = _line("frame-pairs/s, 4-level pyramid with radius-4 lookup (configs[2])", "frame-pairs/s", B / dt, 1, args, 1e3 * dt,
                "configs[2]: 640x480 'stereo' (independent mono streams) + 4-level coarse-to-fine, radius-4 windows, batch %d; extension, no reference counterpart" % B,
                bpp, dict(pairs_per_step=B, levels=pyr.dims, batches_in_flight="back-to-back, one synchronize at the end", pipelined_equals_single=pipelined_equals_single),
                parity_spot_check=parity, roofline=None, cpu_baseline=None)
    pyr.close()
    return out


def workload_loop(args, N, torch, dist, np, synth, world, rank, dev, local_rank):
    """configs[4]; with --gpus N the CANDIDATE SET is sharded (contiguous shards of the key-frame store, the query
    handed to every rank, every rank's best record all-gathered, winner by src/loop_closure.cc:61-65 with ties to the lowest
    global index) -- strong scaling: the 4096 candidates are the job whatever N is."""
    H, W, PD, PC = 480, 640, 720, 480
    NC, MB = args.candidates, 128
    b0, e0 = N.Group.shard(NC, world, rank)
    nloc = e0 - b0
    cf = N.CorrelationFlow(N.default_config(), H, W, max_batch=MB, max_frames=nloc + 1, device=local_rank)
    cv = [synth.canvas(900 + i, H, W) for i in range(8)]
    U = 64
    uniq = np.stack([synth.window(cv[i % 8], H, W, (7 * i) % 120 - 60, (5 * i) % 160 - 80, 0.5 * (i % 11)) for i in range(U)])
    d_uniq = torch.from_numpy(uniq).to(dev)
    for b in range(b0, e0, MB):                                     # candidate i of the global store shows place i % U
        m = min(MB, e0 - b)
        d = d_uniq[torch.arange(b, b + m, device=dev) % U].contiguous()
        cf.intermedium_batch_dev(d.data_ptr(), m, list(range(b - b0, b - b0 + m)))
        cf.synchronize()
    true_idx = 44                                                   # (rotation 0: 44 % 11 == 0) candidates i with i % 64 == 44 hold the query's place
    q = synth.window(cv[true_idx % 8], H, W, (7 * true_idx) % 120 - 60 + 3, (5 * true_idx) % 160 - 80 - 4, 0.0)
    cf.intermedium_u8(q, nloc)                                      # "broadcast the query": every rank computes its spectra
    cf.synchronize()
    cands = list(range(nloc))
    grp, comm, fallback, _ = (None, "nik_group (single GPU: no collective)", False, None) if world == 1 else \
        _make_group(N, torch, dist, np, cf, rank, world, dev, "all-gather of every rank's best record (8 doubles) per query")
    cdev = dev if (world > 1 and dist.get_backend() == "nccl") else "cpu"

    def gather(best, br):
        """this rank's best (local index, result) -> the group's winner (global index, result dict)"""
        if world == 1:
            return best, br
        gi = b0 + best if best >= 0 else -1
        if grp is not None:
            return grp.gather_best([gi], [br])
        rec = torch.zeros(8, dtype=torch.float64)
        if gi >= 0:
            rec[0] = br.info[0] + br.info[1] + br.info[2]; rec[1] = gi
            for k in range(3):
                rec[2 + k] = br.pose[k]; rec[5 + k] = br.info[k]
        else:
            rec[1] = -1
        allr = [torch.zeros(8, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(allr, rec.to(cdev))
        R = np.stack([x.cpu().numpy() for x in allr])
        w = N.Group.pick_best(R)
        return (int(R[w, 1]), dict(pose=R[w, 2:5].tolist(), info=R[w, 5:8].tolist())) if w >= 0 else (-1, None)

    def query(fn):
        best, _, br = fn()
        return gather(best, br)

    def timed(fn, reps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt_own = (time.perf_counter() - t0) / reps
        if world == 1:
            return dt_own, [dt_own]
        t = torch.tensor([dt_own], dtype=torch.float64, device=cdev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        own = [float(x.item()) for x in allt]
        return max(own), own
    reps = max(1, args.steps // 5)
    exact = lambda: query(lambda: cf.match(nloc, cands, raw=True))
    for _ in range(max(1, args.warmup // 2)):
        exact()
    dt, dt_ranks = timed(exact, reps)                                        # exact search, the reference's work per candidate
    dt2 = dt3 = None
    if world == 1:
        dt2, _ = timed(lambda: cf.match_topk(NC, cands, 16), reps)
        cf.set_kzz_cache(True)                                               # key frames are resident: their Kzz can be too
        cf.match(NC, cands, raw=True)
        dt3, _ = timed(lambda: cf.match(NC, cands, raw=True), reps)
        cf.set_kzz_cache(False)
    best, rr, br_raw = cf.match(nloc, cands, raw=True)
    res, br = [rr[i].as_dict() for i in range(nloc)], br_raw.as_dict()
    gbest, gres = gather(best, br_raw)
    if world == 1:
        gres = br
    # size-independent properties at the full size: the winner is the first copy of the true place; every copy of a place scores
    # the same (on every rank); the short-list search finds the same score
    scores = np.array([sum(r["info"]) for r in res])
    first = {k: next((i for i in range(nloc) if (b0 + i) % U == k), None) for k in range(U)}
    same = all(np.all(scores[np.arange(first[k], nloc, U)] == scores[first[k]]) for k in range(U) if first[k] is not None)
    prop = bool(gbest == true_idx and gres is not None and (gres["pose"][0], gres["pose"][1]) == (-4, 3) and same)
    extra = {}
    if world == 1:
        b2, r2, short = cf.match_topk(NC, cands, 16)
        prop = bool(prop and abs(sum(r2["info"]) - sum(br["info"])) < 1e-9 and int(scores.argmax()) == best)
        extra = dict(
            topk16={"candidates_per_s": round(NC / dt2, 1), "ms_per_query": round(1e3 * dt2, 3), "same_best_score": bool(abs(sum(r2["info"]) - sum(br["info"])) < 1e-9),
                    "note": "extension: rank by rotation-stage PSR, full ComputePose on the top 16"},
            kzz_cached_mode={"candidates_per_s": round(NC / dt3, 1), "ms_per_query": round(1e3 * dt3, 3),
                             "bytes_per_candidate": algorithmic_bytes(H, W, PD, PC, kzz_cached=True, hypotheses=2, with_intermedium=False),
                             "frac_of_8TBps": round(NC / dt3 * algorithmic_bytes(H, W, PD, PC, kzz_cached=True, hypotheses=2, with_intermedium=False) / HBM_PEAK, 4),
                             "note": "exact search with the per-keyframe Kzz cache (identical results; the key frames are resident anyway)"})
    if os.environ.get("NIK_BENCH_DUMP"):                            # test hook: the group's winner as this rank sees it
        json.dump(dict(rank=rank, world=world, best=gbest, result=gres, shard=[b0, e0]), open(os.environ["NIK_BENCH_DUMP"] + ".%d" % rank, "w"))
    bpc = algorithmic_bytes(H, W, PD, PC, hypotheses=2, with_intermedium=False)
    out = None
    if rank == 0:
        out = _line("loop-closure candidates/s, exact two-hypothesis ComputePose over resident key frames (configs[4])", "candidates/s", NC / dt, world, args, 1e3 * dt,
                    "configs[4]: 1 query x %d resident key frames (%.1f GB of spectra), not_large_rotation=false on every candidate, strict-> winner" % (NC, NC * 2.62e6 / 1e9),
                    bpc, dict(candidates=NC, chunk=MB, parallelism="candidate set sharded x%d (contiguous shards, lowest global index wins ties)" % world, best_exchange=comm),
                    parity_spot_check=prop, roofline=None, cpu_baseline=None, winner=dict(index=gbest, result=gres),
                    multi_gpu=_multi_gpu_facts(N, world, grp, fallback, candidates_per_rank=[N.Group.shard(NC, world, r)[1] - N.Group.shard(NC, world, r)[0] for r in range(world)],
                                               ms_per_query_per_rank=[round(1e3 * x, 3) for x in dt_ranks],
                                               note="strong scaling: the candidate set is the job; every query ends with one all-gather of 8 doubles per rank"),
                    **extra)
        out["scaling"] = "strong" if world > 1 else "weak"
    if grp is not None:
        grp.close()
    cf.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="pairs", choices=["pairs", "sequence", "pyramid", "hd", "loop4096"])
    ap.add_argument("--batch", type=int, default=512, help="frame pairs per GPU per step (pairs workload; round 6: 512 pairs on 2 streams measured +1.5 ... 3 %% over 256 on 3, profiles/r06_batch_sweep.txt)")
    ap.add_argument("--chunk", type=int, default=0, help="cut every batched call into chunks of at most this many pairs dealt to the streams in turn (nik_set_chunk; 0 = one chunk per stream)")
    ap.add_argument("--streams", type=int, default=0, help="compute streams (lanes) per GPU; 0 = $NIK_STREAMS, else 2 for the pairs workload at >= 384 pairs per step, else the library's 3")
    ap.add_argument("--unique", type=int, default=0, help="distinct synthetic pairs generated (tiled to --batch); 0 = all of them")
    ap.add_argument("--cpu-sample", type=int, default=256, help="pairs timed on the host for cpu_baseline (0 = skip); 256 pairs ~ 25 core-seconds")
    ap.add_argument("--frames", type=int, default=2048, help="sequence workload: frames")
    ap.add_argument("--seq-rot-rate", type=float, default=0.25, help="sequence workload, smooth path: degrees per frame")
    ap.add_argument("--seq-motion", default="sawtooth", choices=["sawtooth", "smooth"], help="sequence workload: synthetic camera path")
    ap.add_argument("--host-frames", action="store_true", help="sequence workload: also run the sequence from HOST memory (pinned and pageable) through nik_tracker_push_host")
    ap.add_argument("--candidates", type=int, default=4096, help="loop4096 workload: resident key frames")
    ap.add_argument("--repeats", type=int, default=0, help="timed regions of --steps steps each (0 = as many as fill --min-time, 3..64); the line reports the median")
    ap.add_argument("--min-time", type=float, default=1.0, help="seconds of timed regions to accumulate when --repeats is 0")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event pass")
    ap.add_argument("--no-cached", action="store_true", help="skip the extra Kzz-cached pass (clean rocprof traces)")
    ap.add_argument("--kernels-dump", default="", help="write the per-kernel HIP-event table to <path>.<pid> (used by the live rocprofv3 passes)")
    ap.add_argument("--no-live-prof", action="store_true", help="skip the live rocprofv3 passes (durations + HBM bytes of the dominant kernel)")
    args = ap.parse_args()
    args.batch_given = any(a == "--batch" or a.startswith("--batch=") for a in sys.argv[1:])

    # `python bench.py --gpus N` on its own starts the N ranks itself (one process per GPU under torch.distributed.run, the
    # launcher the driver uses when it starts the ranks); under a launcher, --gpus must agree with the world it created.
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if env_world is None and args.gpus > 1:
        if args.workload not in ("pairs", "hd", "loop4096"):
            raise SystemExit("--workload %s is a single-GPU measurement" % args.workload)
        port = os.environ.get("MASTER_PORT") or str(29500 + os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.stderr.write("bench.py: --gpus %d without a launcher: starting %d ranks under torch.distributed.run\n" % (args.gpus, args.gpus))
        sys.stderr.flush()
        os.execv(sys.executable, cmd)
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher created WORLD_SIZE=%s ranks: refusing to report a line whose n_gpus is not what was asked for"
                         % (args.gpus, env_world))

    import numpy as np
    import torch
    import torch.distributed as dist
    import synth
    from kcc_helpers import nik
    N = nik()

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

    if args.workload in ("pairs", "hd"):
        out = workload_pairs(args, N, torch, dist, np, synth, world, rank, dev, local_rank, hd=args.workload == "hd")
    elif args.workload == "loop4096":
        out = workload_loop(args, N, torch, dist, np, synth, world, rank, dev, local_rank)
    elif world > 1:
        raise SystemExit("--workload %s is a single-GPU measurement" % args.workload)
    elif args.workload == "sequence":
        out = workload_sequence(args, N, torch, np, synth, dev, local_rank)
    elif args.workload == "pyramid":
        out = workload_pyramid(args, N, torch, np, synth, dev, local_rank)
    else:
        raise SystemExit("unknown workload %s" % args.workload)
    if rank == 0 and out is not None:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
