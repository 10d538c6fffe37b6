# usage (on the GPU box, via gpurun):  bash tools/profile_round.sh r01f
# Produces the round's evidence under gpurun_out/ (copy the files into profiles/ afterwards):
#   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command (1 stream, uncached path)
#   <tag>_pmc_hbm_summary.csv, <tag>_pmc_traffic.json   FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs)
#   <tag>_bench.json         the default bench line (its roofline.traffic comes from the PMC file of this run)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out
CMD="python $R/bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-profile --no-cached"
NIK_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG/stats -- $CMD > $OUT/prof_$TAG.stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  NIK_STREAMS=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/prof_$TAG/pmc_$c -- $CMD > $OUT/prof_$TAG.pmc_$c.log 2>&1
done
python - <<PY
import csv, glob, json, collections, re, shutil, os
out, tag, R = "$OUT", "$TAG", "$R"
st = glob.glob(out + "/prof_%s/stats/**/*kernel_stats.csv" % tag, recursive=True)
if st: shutil.copy(st[0], out + "/%s_kernel_stats.csv" % tag)
AF = {0: "plane", 1: "rot", 3: "u8", 4: "rot8", 5: "polar", 6: "polar", 7: "polar"}; AI = {0: "real", 1: "kernel_fwd", 2: "argmax", 3: "kernel_fwd", 4: "kernel_fwd", 5: "shifted"}
BM = {0: "fwd", 1: "fwd_abs_inv", 2: "mul_inv", 3: "fwd_mul_inv", 4: "solve_inv", 5: "inv", 6: "zz_inv", 7: "mul_inv_x", 8: "fwd_mul_inv_x", 9: "solve_cached"}
def stage(k):
    mu = re.search(r"kA_fwd_u8<(\d+)>", k)
    if mu: return "kA_fwd<%s,u8>" % mu.group(1)
    m = re.search(r"(kA_fwd|kA_inv|kB)<(\d+), (\d+)>", k)
    if not m:
        m2 = re.search(r"kcc::(k_\w+)\(", k); return m2.group(1) if m2 else None
    b, n, mode = m.group(1), int(m.group(2)), int(m.group(3))
    return "%s<%d,%s>" % (b, n, (AF if b == "kA_fwd" else AI if b == "kA_inv" else BM).get(mode, str(mode)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/prof_%s/pmc_%s/**/*counter_collection.csv" % (tag, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c: acc[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
rows, traffic = [], {}
for k, d in sorted(acc.items()):
    if "kcc::" not in k: continue
    f = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1); w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    tot = 2 * f * 1024 + w * 1024          # gfx950: FETCH_SIZE counts half of the bytes read (calibrated)
    s = stage(k)
    rows.append([k, s, len(d["FETCH_SIZE"]), round(f, 1), round(w, 1), round(2 * f * 1024 / 1e6, 1), round(w * 1024 / 1e6, 1), int(tot)])
    if s: traffic[s] = tot
with open(out + "/%s_pmc_hbm_summary.csv" % tag, "w") as o:
    wr = csv.writer(o); wr.writerow(["kernel", "stage", "launches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg", "hbm_read_MB(2x FETCH_SIZE*1024)", "hbm_write_MB(WRITE_SIZE*1024)", "hbm_total_bytes_per_launch"]); wr.writerows(rows)
pm = dict(note="HBM bytes per launch at 256 pairs per launch, 1 stream: 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE counts half the bytes read; calibrated on our own accesses, see DESIGN.md)", pairs_per_launch=256, traffic_bytes_per_launch=traffic)
for dst in (out + "/%s_pmc_traffic.json" % tag, R + "/profiles/%s_pmc_traffic.json" % tag):
    json.dump(pm, open(dst, "w"), indent=1)
print("kernels with traffic:", len(traffic))
PY
cd $R && python bench.py --steps 30 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python -c "
import json; d=json.load(open('$OUT/${TAG}_bench.json')); print(d['value'], d['roofline'], d['cpu_baseline'], d.get('kzz_cached_mode'))"
head -20 $OUT/${TAG}_kernel_stats.csv
