# usage: bash tools/pmc_mfma.sh r01g -- MFMA utilisation of the hot kernels (expected 0: the path has no dense contraction)
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
NIK_STREAMS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma \
  -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-profile --no-cached > $R/gpurun_out/pmc_mfma.log 2>&1
python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$R/gpurun_out/pmc_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kcc::" not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"]][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Kernel_Name"]][r["Counter_Name"]]+=1
names=["SQ_VALU_MFMA_BUSY_CYCLES","SQ_INSTS_VALU_MFMA_MOPS_F32","SQ_BUSY_CYCLES","SQ_INSTS_VALU"]
with open("$R/gpurun_out/${TAG}_pmc_mfma.csv","w") as o:
    w=csv.writer(o); w.writerow(["kernel"]+[n+"_avg_per_launch" for n in names]+["mfma_busy_frac"])
    for k in sorted(acc):
        v=[acc[k][n]/max(cnt[k][n],1) for n in names]
        w.writerow([k]+["%.4g"%x for x in v]+["%.4g"%(v[0]/v[2] if v[2] else 0)])
print(open("$R/gpurun_out/${TAG}_pmc_mfma.csv").read())
PY
