# usage: bash tools/pmc_sq.sh  -- SQ / LDS counter passes for the hot kernels (one rocprofv3 --pmc run per group)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-profile --no-cached"
export NIK_STREAMS=1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES_RESTORED SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcsq/g$i -- $CMD > $R/gpurun_out/pmcsq_g$i.log 2>&1 || echo "group $i failed"
done
python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$R/gpurun_out/pmcsq/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        k=k.replace("void kcc::","").split("(")[0]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
names=sorted({c for k in acc for c in acc[k]})
with open("$R/gpurun_out/pmcsq_summary.csv","w") as o:
    o.write("kernel,"+",".join(names)+"\n")
    for k in acc:
        o.write(k+","+",".join("%.4g"%(acc[k][c]/max(cnt[k][c],1)) for c in names)+"\n")
print(open("$R/gpurun_out/pmcsq_summary.csv").read()[:6000])
PY
