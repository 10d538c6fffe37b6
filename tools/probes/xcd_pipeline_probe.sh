#!/bin/bash
# XCD-resident pipeline probe: builds and runs tools/probes/xcd_pipeline_probe.hip on the GPU box and
# collects the fabric-traffic counters of both forms.  Output: gpurun_out/xcd_probe/*.txt
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/xcd_probe; mkdir -p $OUT
BIN=tools/probes/xcd_pipeline_probe
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN tools/probes/xcd_pipeline_probe.hip || exit 1
export TMPDIR=/tmp
{
echo "== default (calibrated arithmetic, window 2, 5 workgroups per CU)"; timeout 120 $BIN --items 256
for w in 1 3 4 8; do echo "== window $w"; timeout 120 $BIN --items 256 --window $w --mode 2; done
echo "== 4 workgroups per CU"; timeout 120 $BIN --items 256 --wpc 4 --mode 2
echo "== roles 6:5:3:2"; timeout 120 $BIN --items 256 --r1 6 --r2 5 --r3 3 --r4 2 --mode 2
echo "== roles 5:4:4:3"; timeout 120 $BIN --items 256 --r1 5 --r2 4 --r3 4 --r4 3 --mode 2
echo "== no arithmetic (memory only)"; timeout 120 $BIN --items 256 --f1 0 --f2 0 --f3 0 --f4 0
echo "== no arithmetic, window 1"; timeout 120 $BIN --items 256 --f1 0 --f2 0 --f3 0 --f4 0 --window 1 --mode 2
echo "== half arithmetic"; timeout 120 $BIN --items 256 --f1 30 --f2 64 --f3 43 --f4 40
echo "== plain consumer loads (no L1 bypass): the tag check must catch stale words if the protocol needs the bypass"; timeout 120 $BIN --items 256 --bypass 0 --mode 2
echo "== 64 items (intermediates fit the Infinity Cache)"; timeout 120 $BIN --items 64 --reps 40
for dg in "1 1" "2 1" "4 1" "1 2" "2 2" "3 1"; do set -- $dg; echo "== generic workgroups, lag D=$1 G=$2"; timeout 120 $BIN --items 256 --mode 5 --D $1 --G $2; done
echo "== generic, D=1 G=1, 4 workgroups per CU"; timeout 120 $BIN --items 256 --mode 4 --wpc 4
echo "== generic, D=1 G=1, no arithmetic"; timeout 120 $BIN --items 256 --mode 5 --f1 0 --f2 0 --f3 0 --f4 0
echo "== generic, D=2 G=1, no arithmetic"; timeout 120 $BIN --items 256 --mode 5 --D 2 --f1 0 --f2 0 --f3 0 --f4 0
echo "== generic, D=1 G=1, plain consumer loads"; timeout 120 $BIN --items 256 --mode 4 --bypass 0
} > $OUT/runs.txt 2>&1
rm -f $OUT/pmc.txt
# fabric traffic: separate rocprofv3 passes per counter (never combined with other traces)
for c in FETCH_SIZE WRITE_SIZE; do
  for m in 1 2 4 42; do
    rm -rf /tmp/xp_${c}_$m
    extra=""; mm=$m
    if [ $m = 2 ]; then extra="--window 8"; fi
    if [ $m = 42 ]; then extra="--D 2"; mm=4; fi
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/xp_${c}_$m -- $OLDPWD/$BIN --items 256 --reps 4 --mode $mm $extra > /dev/null 2>&1)
    python3 - "$c" "$m" /tmp/xp_${c}_$m >> $OUT/pmc.txt <<'PY'
import csv, glob, sys, collections
c, m, d = sys.argv[1:4]
tot = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == c:
            k = r["Kernel_Name"].split("(")[0]
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in sorted(tot):
    print("%s mode %s  %-40s launches %3d  %s per launch = %.1f (KB units: %.1f MB)" % (c, m, k[:40], cnt[k], c, tot[k] / cnt[k], tot[k] / cnt[k] / 1024.0))
PY
  done
done
cat $OUT/runs.txt $OUT/pmc.txt
