# usage: bash tools/batch_sweep.sh  -- throughput and per-kernel GB/s against pairs per launch (Infinity Cache residency)
cd $GRAFT_REPO_ROOT
for b in 16 32 48 64 96 128 256; do
  for s in 2 4; do
    NIK_STREAMS=$s python bench.py --batch $b --steps $((7680/b)) --warmup 4 --cpu-sample 0 --no-cached > gpurun_out/bs_${b}_$s.json 2>gpurun_out/bs_${b}_$s.err || echo "FAIL $b $s"
  done
done
python - <<PY
import json
for b in (16,32,48,64,96,128,256):
    row=[]
    for s in (2,4):
        try: d=json.load(open("gpurun_out/bs_%d_%d.json"%(b,s))); row.append("%8.0f"%d["value"])
        except Exception as e: row.append("  ERR")
    d=json.load(open("gpurun_out/bs_%d_2.json"%b))
    tot=sum(k["avg_ms"] for k in d["kernels"])
    print(b, " ".join(row), "sum_kernels_ms/pair %.5f"%(tot/b), " ".join("%s=%.0f"%(k["name"].replace("kernel_fwd","kf")[:18],k["gbps"]) for k in d["kernels"][:9]))
PY
