#!/usr/bin/env python
"""first_8gpu.py -- the first multi-GPU run of this library, scripted so that it cannot surprise.

For every workload that shards (pairs, hd, loop4096) and every N in --gpus (default 1 2 4 8) it runs
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
(the driver's own launch form; N = 1 also as plain `python bench.py`) and ASSERTS, from the JSON line alone:
  * n_gpus == N and multi_gpu.world == N;
  * N > 1: multi_gpu.rccl_ranks == N (the library's OWN RCCL communicator spans all ranks -- reference semantics being sharded:
    the shared CorrelationFlow of map_builder.cc:23-26, the winner rule of loop_closure.cc:61-65), multi_gpu.fallback is false;
  * pairs / hd: the slowest and the fastest rank of the median region within 10 % of each other;
  * loop4096: every rank named the same winner (parity_spot_check) and the shards add up to the candidate set;
  * N = 1 under the launcher equals the plain single-GPU line within 3 % (the launcher costs nothing).
It prints one table (value, per-GPU value, efficiency vs N = 1) and exits non-zero on the first violated assertion, after writing
every line it got to --out.

--selftest: what a ONE-GPU box can exercise of this script -- N = 1 for the three workloads, plus the 2-rank path with both ranks
on device 0 and gloo for the rendezvous (bench.py's NIK_BENCH_DEVICE / NIK_BENCH_BACKEND hooks; a 2-rank RCCL communicator cannot
form on one GPU, so rccl_ranks is expected to be 0 there and the per-rank balance is not asserted).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(workload, n, launcher, extra, env_extra=None, timeout=1200):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    common = [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", workload] + extra
    if launcher:
        port = 29500 + (os.getpid() + int(time.time())) % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + common
    else:
        cmd = [sys.executable] + common
    t0 = time.time()
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        raise RuntimeError("%s --gpus %d failed (rc %d):\n%s\n%s" % (workload, n, p.returncode, p.stdout[-1500:], p.stderr[-3000:]))
    j = json.loads(lines[-1])
    j["_wall_s"] = round(time.time() - t0, 1)
    return j


def check(j, workload, n, selftest_gloo=False):
    """the assertions of the module docstring on one line; returns a list of violations (empty: fine)"""
    bad = []
    mg = j.get("multi_gpu") or {}
    if j.get("n_gpus") != n:
        bad.append("n_gpus %s != %d" % (j.get("n_gpus"), n))
    if mg.get("world") != n:
        bad.append("multi_gpu.world %s != %d" % (mg.get("world"), n))
    if n > 1 and not selftest_gloo:
        if mg.get("rccl_ranks") != n:
            bad.append("multi_gpu.rccl_ranks %s != %d (no RCCL communicator over all ranks)" % (mg.get("rccl_ranks"), n))
        if mg.get("fallback") is not False:
            bad.append("multi_gpu.fallback is %s (the exchange ran through torch.distributed, not nik_group)" % mg.get("fallback"))
    if workload in ("pairs", "hd"):
        lo, hi = mg.get("pairs_per_s_per_rank_min"), mg.get("pairs_per_s_per_rank_max")
        if not (lo and hi and lo > 0):
            bad.append("per-rank rates missing")
        elif n > 1 and not selftest_gloo and hi / lo > 1.10:
            bad.append("ranks out of balance: %.0f .. %.0f pairs/s (%.1f %%)" % (lo, hi, 100 * (hi / lo - 1)))
    if workload == "loop4096":
        if j.get("parity_spot_check") is not True:
            bad.append("loop4096: the ranks did not agree on the winner (parity_spot_check %s)" % j.get("parity_spot_check"))
        cpr = mg.get("candidates_per_rank") or []
        if sum(cpr) != (j.get("config") or {}).get("candidates"):
            bad.append("candidate shards %s do not add up to %s" % (cpr, (j.get("config") or {}).get("candidates")))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--workloads", nargs="*", default=["pairs", "hd", "loop4096"])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "first_8gpu.json"))
    ap.add_argument("--selftest", action="store_true", help="one-GPU box: N = 1 and the 2-rank one-device gloo hook")
    ap.add_argument("--quick", action="store_true", help="short runs (script check, not a measurement)")
    args = ap.parse_args()
    extra = ["--cpu-sample", "0", "--no-live-prof", "--no-cached", "--no-profile"] + (["--steps", "3", "--warmup", "1", "--repeats", "2"] if args.quick else [])
    per_wl = {"pairs": [], "hd": [], "loop4096": ["--candidates", "512"] if args.quick or args.selftest else []}
    results, failures = [], []

    def record(tag, wl, n, j, bad):
        results.append(dict(tag=tag, workload=wl, n_gpus=n, value=j.get("value"), unit=j.get("unit"), ms_per_step=j.get("ms_per_step"),
                            multi_gpu=j.get("multi_gpu"), frac_of_8TBps=(j.get("path_roofline") or {}).get("frac_of_8TBps"), wall_s=j.get("_wall_s"),
                            violations=bad))
        for b in bad:
            failures.append("%s %s --gpus %d: %s" % (tag, wl, n, b))

    for wl in args.workloads:
        base = None
        plain = run(wl, 1, False, extra + per_wl[wl])
        record("plain", wl, 1, plain, check(plain, wl, 1))
        ns = [1] if args.selftest else args.gpus
        for n in ns:
            j = run(wl, n, True, extra + per_wl[wl])
            bad = check(j, wl, n)
            if n == 1:
                base = j["value"]
                if abs(j["value"] / plain["value"] - 1.0) > 0.03:
                    bad.append("N = 1 under the launcher %.1f vs plain %.1f: more than 3 %% apart" % (j["value"], plain["value"]))
            record("launcher", wl, n, j, bad)
        if args.selftest:
            # two ranks on device 0, gloo rendezvous: exercises the multi-rank code of bench.py and of this script
            j = run(wl, 2, True, extra + per_wl[wl], dict(NIK_BENCH_DEVICE="0", NIK_BENCH_BACKEND="gloo"))
            bad = check(j, wl, 2, selftest_gloo=True)
            if (j.get("multi_gpu") or {}).get("rccl_ranks") not in (0, None):
                bad.append("gloo hook: rccl_ranks %s, expected 0" % j["multi_gpu"]["rccl_ranks"])
            record("selftest-2-ranks-one-device-gloo", wl, 2, j, bad)
        del base
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(dict(results=results, failures=failures), open(args.out, "w"), indent=1)
    print("%-34s %-9s %3s %12s %12s %8s %8s" % ("run", "workload", "N", "value", "per GPU", "eff", "frac"))
    one = {}
    for r in results:
        if r["tag"] == "launcher" and r["n_gpus"] == 1:
            one[r["workload"]] = r["value"]
    for r in results:
        eff = r["value"] / (one.get(r["workload"], r["value"]) * (r["n_gpus"] if r["workload"] != "loop4096" else r["n_gpus"])) if r["tag"] == "launcher" else float("nan")
        print("%-34s %-9s %3d %12.1f %12.1f %8.3f %8s  %s" % (r["tag"], r["workload"], r["n_gpus"], r["value"], r["value"] / r["n_gpus"], eff,
                                                             r["frac_of_8TBps"], "OK" if not r["violations"] else "VIOLATED: " + "; ".join(r["violations"])))
    if failures:
        print("\n".join(["FAILED:"] + failures))
        sys.exit(1)
    print("first_8gpu: all assertions hold")


if __name__ == "__main__":
    main()
