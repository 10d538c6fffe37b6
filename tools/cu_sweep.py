#!/usr/bin/env python
"""cu_sweep.py -- VERDICT r5 item 1(a): what do the kernels lose on fewer CUs, and does a spatial partition of the chip
(hipExtStreamCreateWithCUMask, $NIK_LANE_CUS) beat time-sharing it?

Part 1: every kernel of the headline step timed alone (HIP events, one stream, 256 pairs) on the first K CUs, K = 256 .. 64.
Part 2: the headline step itself with the lanes as plain streams (today) and as disjoint CU partitions.
All on one box, back to back.  Usage: python tools/cu_sweep.py [out.txt]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(env_extra, extra_args=()):
    env = dict(os.environ)
    env.update(env_extra)
    env["NIK_LIB"] = os.path.join(ROOT, "ni-slam_amd", "libnislam_kcc_hip_tune.so")     # $NIK_LANE_CUS is a laboratory switch
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-sample", "0", "--no-live-prof", "--no-cached", "--steps", "20", "--warmup", "3"] + list(extra_args)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    sys.stderr.write(r.stderr[-2000:])
    return None


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout

    def say(s):
        out.write(s + "\n"); out.flush()
        if out is not sys.stdout:
            print(s, flush=True)

    say("# part 1: per-kernel ms per 256 pairs (HIP events, one stream) on the first K CUs")
    ks = [256, 224, 192, 160, 128, 96, 64]
    tab, names = {}, []
    for k in ks:
        j = bench({} if k == 256 else {"NIK_LANE_CUS": str(k)}, ["--repeats", "1"])
        if not j:
            say("K=%d failed" % k); continue
        tab[k] = {x["name"]: x["avg_ms"] for x in j["kernels"]}
        if not names:
            names = [x["name"] for x in j["kernels"]]
        say("K=%3d: one-stream-set value %.1f pairs/s (3 lanes on the same K CUs)" % (k, j["value"]))
    say("%-28s" % "kernel" + "".join("%9d" % k for k in ks if k in tab) + "   t(128)/t(256)")
    for n in names:
        row = "%-28s" % n + "".join("%9.4f" % tab[k].get(n, float("nan")) for k in ks if k in tab)
        if 128 in tab and 256 in tab and n in tab[128] and n in tab[256]:
            row += "   %.2f" % (tab[128][n] / tab[256][n])
        say(row)
    say("%-28s" % "sum" + "".join("%9.4f" % sum(tab[k].values()) for k in ks if k in tab))

    say("\n# part 2: the headline step (256 pairs), lanes as plain streams vs disjoint CU partitions; pairs/s median [min, max]")
    cases = [("1 stream", {"NIK_STREAMS": "1"}),
             ("2 streams", {"NIK_STREAMS": "2"}),
             ("2 partitions 128+128", {"NIK_STREAMS": "2", "NIK_LANE_CUS": "128,128"}),
             ("3 streams (release default)", {"NIK_STREAMS": "3"}),
             ("3 partitions 88+88+80", {"NIK_STREAMS": "3", "NIK_LANE_CUS": "88,88,80"}),
             ("4 streams", {"NIK_STREAMS": "4"}),
             ("4 partitions 64x4", {"NIK_STREAMS": "4", "NIK_LANE_CUS": "64,64,64,64"}),
             ("2 streams, batch 512", {"NIK_STREAMS": "2"}, ["--batch", "512"]),
             ("2 partitions, batch 512", {"NIK_STREAMS": "2", "NIK_LANE_CUS": "128,128"}, ["--batch", "512"]),
             ("3 streams, batch 512", {"NIK_STREAMS": "3"}, ["--batch", "512"]),
             ("3 streams (again)", {"NIK_STREAMS": "3"})]
    for c in cases:
        name, env = c[0], c[1]
        extra = list(c[2]) if len(c) > 2 else []
        j = bench(env, ["--no-profile"] + extra)
        if not j:
            say("%-32s failed" % name); continue
        t = j["timing"]
        say("%-32s %9.1f  [%9.1f, %9.1f]  regions %d" % (name, j["value"], t["value_min"], t["value_max"], t["regions"]))


if __name__ == "__main__":
    main()
