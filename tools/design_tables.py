#!/usr/bin/env python
"""design_tables.py -- prints the live tables of DESIGN.md from the committed evidence files of a round (profiles/<tag>_*), so
that every number in the document is one a reader can find in a file.  usage: python tools/design_tables.py r06"""
import csv
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = lambda n: os.path.join(R, "profiles", n)      # noqa: E731


def kernels_table(bench_file, traffic_file, sq_file=None):
    b = json.load(open(P(bench_file)))
    B = b["config"]["pairs_per_gpu_per_step"]
    tr = json.load(open(P(traffic_file)))["traffic_bytes_per_launch"] if os.path.exists(P(traffic_file)) else {}
    sq = {}
    if sq_file and os.path.exists(P(sq_file)):
        for r in csv.DictReader(open(P(sq_file))):
            sq[r["stage"]] = r
    print("| kernel | ms per %d items (HIP events, 1 stream) | design MB | moved MB (L2 fills + writes) | GB/s on design bytes | nominal MB (SURVEY 8d planes) | VALU busy | LDS busy | LDS conflict share |" % B)
    print("|---|---|---|---|---|---|---|---|---|")
    tot = dict(ms=0.0, d=0.0, m=0.0, n=0.0)
    for k in b["kernels"]:
        n = k["name"]; s = sq.get(n, {})
        mv = tr.get(n)
        print("| `%s` | %.4f | %.0f | %s | %.0f | %.0f | %s | %s | %s |" % (
            n, k["avg_ms"], k["design_bytes_per_launch"] / 1e6, "%.0f" % (mv / 1e6) if mv else "—", k["gbps"], k["bytes_per_launch"] / 1e6,
            s.get("valu_busy_share_of_simd_cycles", "—"), s.get("lds_busy_share", "—"), s.get("lds_bank_conflict_share_of_lds_active", "—")))
        tot["ms"] += k["avg_ms"]; tot["d"] += k["design_bytes_per_launch"]; tot["n"] += k["bytes_per_launch"]; tot["m"] += mv or k["design_bytes_per_launch"]
    print("| **sum** | **%.3f** (step %.3f) | %.0f = %.2f MB/pair | %.0f = %.2f MB/pair | %.0f | %.0f = %.2f MB/pair | | | |" % (
        tot["ms"], b["ms_per_step"], tot["d"] / 1e6, tot["d"] / B / 1e6, tot["m"] / 1e6, tot["m"] / B / 1e6, tot["d"] / tot["ms"] / 1e6, tot["n"] / 1e6, tot["n"] / B / 1e6))
    return b


print("### headline (profiles/%s_bench.json, %s_pmc_traffic.json, %s_pmc_sq_table.csv)\n" % (tag, tag, tag))
b = kernels_table(tag + "_bench.json", tag + "_pmc_traffic.json", tag + "_pmc_sq_table.csv")
print("\nvalue %.1f pairs/s [%s .. %s], ms/step %.4f, path frac %.4f, roofline %s\ncpu %s\nparity %s\nkzz cached %s" % (
    b["value"], b["timing"]["value_min"], b["timing"]["value_max"], b["ms_per_step"], b["path_roofline"]["frac_of_8TBps"],
    json.dumps({k: b["roofline"][k] for k in ("kernel", "achieved", "frac", "frac_design", "frac_moved_bytes", "frac_contract", "nominal_bytes_over_contract", "avg_ms_hip_event", "avg_ms_rocprof", "durations_agree_within_5pct", "traffic")}),
    json.dumps(b["cpu_baseline"]), json.dumps({k: v for k, v in b["parity_spot_check"].items() if k != "note"}), json.dumps(b.get("kzz_cached_mode"))))
print("\n### HD (profiles/%s_hd_bench.json)\n" % tag)
h = kernels_table(tag + "_hd_bench.json", tag + "_hd_pmc_traffic.json")
print("\nvalue %.1f, frac %.4f" % (h["value"], h["path_roofline"]["frac_of_8TBps"]))
print("\n### other workloads")
for w in ("sequence", "pyramid", "loop4096"):
    d = json.load(open(P("%s_workload_%s.json" % (tag, w))))
    print(w, d["value"], d["unit"], d["path_roofline"]["frac_of_8TBps"], json.dumps(d.get("host_inclusive")), json.dumps(d.get("kzz_cached_mode"))[:300], json.dumps(d.get("topk16"))[:200])
lat = json.load(open(P(tag + "_latency.json")))
print("latency", json.dumps(lat)[:600])
for f in (tag + "_parity_sweep.json", tag + "_parity_sweep_hd.json"):
    d = json.load(open(P(f)))
    for m in ("small_rot", "large_rot"):
        r = d[m]
        print(f, m, {k: r[k] for k in r if k not in ("translation_near_ties(gap,pixels)", "first_failures", "theta_difference_examples")})
