# usage (GPU box): bash tools/fetch_calib.sh  -- calibration of the FETCH_SIZE counter on this library's access patterns
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/fetch_calib
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/fetch_calib -- $R/tools/probes/fetch_calib > $R/gpurun_out/fetch_calib.log 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/fetch_calib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE": acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
GiB = 1 << 30
want = {"k_lane16": GiB, "k_lane8": GiB, "k_lane4": GiB, "void k_chunk64<256>": GiB // 4, "void k_chunk64<64>": GiB}
print(open("$R/gpurun_out/fetch_calib.log").read().strip().splitlines()[-1])
for k, v in sorted(acc.items()):
    w = want.get(k)
    print("%-22s FETCH_SIZE %12.0f KB  = %.3f x the %s bytes read" % (k, v[0], v[0] * 1024 / w, w) if w else (k, v))
PY
