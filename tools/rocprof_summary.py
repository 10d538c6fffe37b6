"""rocprofv3 passes over a bench command and their per-kernel summaries, shared by bench.py (live cross-check of its
HIP-event timings and live HBM traffic of the dominant kernel) and tools/profile_round.sh (the round's committed evidence).

Three separate runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes: `--kernel-trace --stats` for durations, and one
`--pmc` pass each for FETCH_SIZE and WRITE_SIZE (they do not fit one pass; counters are never combined with sys/runtime
traces).  gfx950 correction: FETCH_SIZE counts half of the bytes of wide coalesced reads (calibrated on this library's own
accesses), so HBM bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

AF = {0: "plane", 1: "rot", 3: "u8", 4: "rot8", 5: "polar", 6: "polar", 7: "polar"}
AI = {0: "real", 1: "kernel_fwd", 2: "argmax", 3: "kernel_fwd", 4: "kernel_fwd", 5: "shifted", 6: "argmax_win"}
BM = {0: "fwd", 1: "fwd_abs_inv", 2: "mul_inv", 3: "fwd_mul_inv", 4: "solve_inv", 5: "inv", 6: "zz_inv", 7: "mul_inv_x",
      8: "fwd_mul_inv_x", 9: "solve_cached"}


def stage(kernel_name):
    """demangled kernel name -> the stage name bench.py's HIP-event profiler uses (kB<480,fwd_mul_inv> ...)"""
    mu = re.search(r"kA_fwd_u8<(\d+)>", kernel_name)
    if mu:
        return "kA_fwd<%s,u8>" % mu.group(1)
    m = re.search(r"(kA_fwd|kA_inv|kB)<(\d+), (\d+)", kernel_name)
    if not m:
        m2 = re.search(r"kcc::(k_\w+)\(", kernel_name)
        return m2.group(1) if m2 else None
    b, n, mode = m.group(1), int(m.group(2)), int(m.group(3))
    return "%s<%d,%s>" % (b, n, (AF if b == "kA_fwd" else AI if b == "kA_inv" else BM).get(mode, str(mode)))


def _run(cmd, env, log, timeout):
    with open(log, "w") as lf:
        return subprocess.run(cmd, env=env, stdout=lf, stderr=subprocess.STDOUT, timeout=timeout, cwd="/tmp").returncode


def collect(bench_cmd, outdir, want_pmc=True, timeout=240, extra_env=None):
    """run the passes; returns {"stats": {stage: {avg_ms, min_ms, calls}}, "traffic": {stage: bytes per launch},
    "pmc": {stage: {fetch_kb, write_kb, launches}}, "errors": [...]}"""
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp", NIK_STREAMS="1")
    env.update(extra_env or {})
    res = dict(stats={}, traffic={}, pmc={}, errors=[], files={})
    passes = [("stats", ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(outdir, "stats"), "--"])]
    if want_pmc:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            passes.append((c, ["rocprofv3", "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(outdir, "pmc_" + c), "--"]))
    for name, head in passes:
        try:
            shutil.rmtree(os.path.join(outdir, "stats" if name == "stats" else "pmc_" + name), ignore_errors=True)
            rc = _run(head + list(bench_cmd), env, os.path.join(outdir, name + ".log"), timeout)
            if rc != 0:
                res["errors"].append("%s pass: rc %d" % (name, rc))
        except Exception as e:                                  # noqa: BLE001 -- a profiler failure must never take the bench down
            res["errors"].append("%s pass: %s" % (name, str(e)[:200]))
    st = glob.glob(os.path.join(outdir, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        res["files"]["kernel_stats"] = st[0]
        for r in csv.DictReader(open(st[0])):
            s = stage(r["Name"])
            if s and "kcc::" in r["Name"]:
                res["stats"][s] = dict(avg_ms=float(r["AverageNs"]) * 1e-6, min_ms=float(r["MinNs"]) * 1e-6, calls=int(r["Calls"]), kernel=r["Name"])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(outdir, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c:
                    acc[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        s = stage(k)
        if not s or "kcc::" not in k or not d["FETCH_SIZE"] or not d["WRITE_SIZE"]:
            continue
        f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        res["pmc"][s] = dict(kernel=k, launches=len(d["FETCH_SIZE"]), fetch_kb=f, write_kb=w)
        res["traffic"][s] = 2 * f * 1024 + w * 1024
    return res


def write_round_files(res, out, tag, pairs_per_launch):
    """the committed evidence: <tag>_kernel_stats.csv (rocprof's own file), <tag>_kernel_times.json, <tag>_pmc_hbm_summary.csv,
    <tag>_pmc_traffic.json"""
    if res["files"].get("kernel_stats"):
        shutil.copy(res["files"]["kernel_stats"], os.path.join(out, "%s_kernel_stats.csv" % tag))
    json.dump(dict(note="rocprofv3 --kernel-trace --stats of the bench command, one stream; ms per launch of %d pairs" % pairs_per_launch,
                   pairs_per_launch=pairs_per_launch, kernels={s: {k: (round(v, 6) if isinstance(v, float) else v) for k, v in d.items()}
                                                               for s, d in sorted(res["stats"].items())}),
              open(os.path.join(out, "%s_kernel_times.json" % tag), "w"), indent=1)
    with open(os.path.join(out, "%s_pmc_hbm_summary.csv" % tag), "w") as o:
        wr = csv.writer(o)
        wr.writerow(["kernel", "stage", "launches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg", "hbm_read_MB(2x FETCH_SIZE*1024)",
                     "hbm_write_MB(WRITE_SIZE*1024)", "hbm_total_bytes_per_launch"])
        for s, d in sorted(res["pmc"].items()):
            wr.writerow([d["kernel"], s, d["launches"], round(d["fetch_kb"], 1), round(d["write_kb"], 1), round(2 * d["fetch_kb"] * 1024 / 1e6, 1),
                         round(d["write_kb"] * 1024 / 1e6, 1), int(res["traffic"][s])])
    json.dump(dict(note="HBM bytes per launch at %d pairs per launch, 1 stream: 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE counts "
                        "half the bytes read; calibrated on our own accesses, see DESIGN.md)" % pairs_per_launch,
                   pairs_per_launch=pairs_per_launch, traffic_bytes_per_launch=res["traffic"]),
              open(os.path.join(out, "%s_pmc_traffic.json" % tag), "w"), indent=1)


if __name__ == "__main__":
    # python tools/rocprof_summary.py <tag> <outdir> <pairs_per_launch> -- <bench command ...>
    tag, out, ppl = sys.argv[1], sys.argv[2], int(sys.argv[3])
    cmd = sys.argv[sys.argv.index("--") + 1:]
    r = collect(cmd, os.path.join(out, "prof_" + tag))
    write_round_files(r, out, tag, ppl)
    print("kernels timed:", len(r["stats"]), "with traffic:", len(r["traffic"]), "errors:", r["errors"])
