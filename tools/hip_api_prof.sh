# usage (GPU box): WL=pyramid bash tools/hip_api_prof.sh  -- host-side HIP API time per call type for one workload
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --stats --output-format csv -d $R/gpurun_out/hipapi -- python $R/bench.py --workload ${WL:-pyramid} --steps 200 --warmup 10 --cpu-sample 0 > $R/gpurun_out/hipapi.log 2>&1
grep '"metric"' $R/gpurun_out/hipapi.log | cut -c80-140
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/hipapi/**/*hip_api_stats.csv",recursive=True)
print(f)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:14]: print(r["Name"][:40], r["Calls"], "tot ms", float(r["TotalDurationNs"])/1e6, "avg us", float(r["AverageNs"])/1e3)
PY
