cd $GRAFT_REPO_ROOT
for v in "" _p360_15x24 _p360_20x18 _p360_24x15 _p240_16x15 _p240_12x20 _p480_20x24 _p480_6x8x10 _p480_10x6x8 _p640_10x8x8 _p640_20x32 _p640_16x40; do
  NIK_LIB=$PWD/ni-slam_amd/libnislam_kcc_hip$v.so python bench.py --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/var$v.json 2>gpurun_out/var$v.err || echo "FAIL $v"
done
python - <<PY
import json,glob,os
base=json.load(open("gpurun_out/var.json"))
bk={k["name"]:k["avg_ms"] for k in base["kernels"]}
print("base", base["value"], base["parity_spot_check"])
for f in sorted(glob.glob("gpurun_out/var_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f,"ERR"); continue
    diffs=["%s %.3f->%.3f"%(k["name"],bk.get(k["name"],0),k["avg_ms"]) for k in d["kernels"] if abs(k["avg_ms"]-bk.get(k["name"],0))>0.008]
    print(os.path.basename(f), d["value"], d["parity_spot_check"], "; ".join(diffs))
PY
