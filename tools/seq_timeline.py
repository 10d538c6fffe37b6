"""usage: python tools/seq_timeline.py <rocprofv3 sqlite db> -- where the sequence workload's time goes.

From a --kernel-trace of `bench.py --workload sequence` (tools/seq_trace.sh): the longest timed run is cut into the three kinds
of work the tracker enqueues -- ComputeIntermedium of a window (kA_fwd_u8, kB fwd_abs_inv, kA_inv shifted, kA_fwd polar, kB fwd),
the per-keyframe Kzz kernels of the cache (kB zz_inv, kA_inv kernel_fwd on one plane, kB fwd, k_store_mzz) and the look-ahead
ComputePose batches -- and for each: launches, summed duration, workgroups per launch against the chip's resident slots, and
what the same work costs per item at batch 256 (profiles/r06_bench.json kernels / 256).  Plus: GPU-busy union, mean number of
kernels running, idle gaps, per-stream busy time."""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name, grid_x, grid_y, grid_z, workgroup_x, stream_id, lds_size from kernels order by start").fetchall()
runs, cur, ce = [], [rows[0]], rows[0][1]
for r in rows[1:]:
    if r[0] - ce > 1.5e6:
        runs.append(cur); cur = []
    cur.append(r); ce = max(ce, r[1])
runs.append(cur)
run = max(runs, key=len)
t0, t1 = run[0][0], max(r[1] for r in run)
span = (t1 - t0) / 1e6


def kind(name):
    n = re.sub(r"\(.*", "", name)
    n = n.replace("void kcc::", "").replace("kcc::", "")
    return n


MODE_B = {0: "fwd", 1: "fwd_abs_inv", 2: "mul_inv", 3: "fwd_mul_inv", 4: "solve_inv", 5: "inv", 6: "zz_inv", 7: "mul_inv_x", 8: "fwd_mul_inv_x", 9: "solve_cached"}
EPI_A = {0: "real", 1: "kernel_fwd", 2: "argmax", 3: "kernel_fwd(n)", 4: "kernel_fwd(gauss)", 5: "shifted", 6: "argmax_win"}
SRC_A = {0: "plane", 1: "rot", 3: "u8", 4: "rot8", 5: "polar", 6: "polar", 7: "polar"}


def pretty(n):
    m = re.match(r"kB<(\d+), (\d+)>", n)
    if m:
        return "kB<%s,%s>" % (m.group(1), MODE_B.get(int(m.group(2)), m.group(2)))
    m = re.match(r"kA_inv<(\d+), (\d+), (\d+)>", n)
    if m:
        return "kA_inv<%s,%s>" % (m.group(1), EPI_A.get(int(m.group(2)), m.group(2)))
    m = re.match(r"kA_fwd<(\d+), (\d+)>", n)
    if m:
        return "kA_fwd<%s,%s>" % (m.group(1), SRC_A.get(int(m.group(2)), m.group(2)))
    return n


def group(p):
    if p.startswith(("kA_fwd_u8", "kB<640,fwd_abs_inv>", "kA_inv<240,shifted>", "kA_fwd<360,polar>")):
        return "intermedium"
    if "zz_inv" in p or p.startswith("k_store_mzz"):
        return "kzz (per key frame)"
    if p in ("kB<480,fwd>", "kB<640,fwd>"):
        return "intermedium / kzz (B forward)"
    if p.startswith(("__amd", "k_finalize", "k_residual")):
        return "small (finalize, copies)"
    return "pose batches"


agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
gagg = collections.defaultdict(lambda: [0, 0.0])
streams = collections.defaultdict(float)
for a, b, n, gx, gy, gz, wx, sid, lds in run:
    p = pretty(kind(n))
    wgs = (gx // max(wx, 1)) * max(gy, 1) * max(gz, 1)
    agg[p][0] += 1; agg[p][1] += (b - a) / 1e3; agg[p][2] += wgs
    g = group(p); gagg[g][0] += 1; gagg[g][1] += (b - a) / 1e3
    streams[sid] += (b - a) / 1e6
ev = sorted((a, b) for a, b, *_ in run)
busy, gaps, (cs, ce) = 0, [], ev[0]
for a, b in ev[1:]:
    if a > ce:
        busy += ce - cs; gaps.append(a - ce); cs, ce = a, b
    else:
        ce = max(ce, b)
busy += ce - cs
tot = sum(b - a for a, b in ev) / 1e6
print("longest timed run: %d kernels, span %.2f ms, GPU busy (union) %.2f ms = %.0f %%, sum of kernel durations %.2f ms -> %.2f kernels running on average while busy"
      % (len(run), span, busy / 1e6, 100 * busy / 1e6 / span, tot, tot / (busy / 1e6)))
big = [g for g in gaps if g > 2e4]
print("idle: %.2f ms in %d gaps > 20 us, %.2f ms in gaps of 5-20 us, %.2f ms in shorter ones" % (
    sum(big) / 1e6, len(big), sum(g for g in gaps if 5e3 < g <= 2e4) / 1e6, sum(g for g in gaps if g <= 5e3) / 1e6))
print("per stream busy (sum of durations, ms): " + ", ".join("%s: %.2f" % (k, v) for k, v in sorted(streams.items(), key=lambda kv: -kv[1])))
print("\nby kind of work (sum of kernel durations):")
for g, (c, t) in sorted(gagg.items(), key=lambda kv: -kv[1][1]):
    print("  %-32s %5d launches %8.2f ms  %4.1f %%" % (g, c, t / 1e3, 100 * t / 1e3 / tot))
print("\nby kernel: launches, total, average, average workgroups per launch")
for p, (c, t, w) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-30s %5d  %8.1f us  avg %6.1f us  %7.0f workgroups" % (p, c, t, t / c, w / c))
