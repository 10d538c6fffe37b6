"""usage: python tools/seq_trace.py <rocprofv3 sqlite db>  -- the sequence workload's timed runs in a kernel trace: span, GPU-busy
union, idle gaps by size, and per-kernel totals of the longest run (tools/seq_trace.sh)"""
import collections
import re
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name, grid_x, workgroup_x from kernels order by start").fetchall()
runs, cur, ce = [], [rows[0]], rows[0][1]
for r in rows[1:]:
    if r[0] - ce > float(__import__("os").environ.get("GAP", "1.5e6")):
        runs.append(cur); cur = []
    cur.append(r); ce = max(ce, r[1])
runs.append(cur)
runs = [s for s in runs if len(s) > 1000]
for s in runs:
    ev = [(a, b) for a, b, *_ in s]
    t0, t1 = ev[0][0], max(b for a, b in ev)
    busy, gaps, (cs, ce) = 0, [], ev[0]
    for a, b in ev[1:]:
        if a > ce:
            busy += ce - cs; gaps.append(a - ce); cs, ce = a, b
        else:
            ce = max(ce, b)
    busy += ce - cs
    g = np.array(gaps)
    print("run: %5d kernels, span %.2f ms, busy %.2f ms (%.0f %%), sum of durations %.2f ms; idle gaps > 20 us: %d = %.2f ms, 5-20 us: %d = %.2f ms, < 5 us: %.2f ms"
          % (len(s), (t1 - t0) / 1e6, busy / 1e6, 100 * busy / (t1 - t0), sum(b - a for a, b in ev) / 1e6, (g > 2e4).sum(), g[g > 2e4].sum() / 1e6,
             ((g > 5e3) & (g <= 2e4)).sum(), g[(g > 5e3) & (g <= 2e4)].sum() / 1e6, g[g <= 5e3].sum() / 1e6))
s = max(runs, key=len)
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for a, b, n, gx, wx in s:
    n = re.sub(r"\(.*", "", n)[:60]
    agg[n][0] += 1; agg[n][1] += (b - a) / 1e3; agg[n][2] += gx // max(wx, 1)
print("longest run by kernel:")
for n, (c, t, wg) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-60s calls %4d  total %8.1f us  avg %6.1f us  avg workgroups(x) %6.0f" % (n, c, t, t / c, wg / c))
