# usage: bash tools/runvar_hd.sh <suffix> ...   -- the hd workload (1280x720) on tuning variants of the library
cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  NIK_LIB=$PWD/ni-slam_amd/libnislam_kcc_hip$v.so python bench.py --workload hd --steps ${STEPS:-20} --warmup 3 --cpu-sample 4 --no-live-prof > gpurun_out/hdvar$v.json 2>gpurun_out/hdvar$v.err || echo "FAIL $v"
done
python - "$@" <<PY
import json,sys
base=json.load(open("gpurun_out/hdvar.json"))
bk={k["name"]:k["avg_ms"] for k in base["kernels"]}
print("base", base["value"], base["parity_spot_check"])
for v in sys.argv[1:]:
    try: d=json.load(open("gpurun_out/hdvar%s.json"%v))
    except Exception as e: print(v,"ERR", open("gpurun_out/hdvar%s.err"%v).read()[-300:]); continue
    diffs=["%s %.3f->%.3f"%(k["name"],bk.get(k["name"],0),k["avg_ms"]) for k in d["kernels"] if abs(k["avg_ms"]-bk.get(k["name"],0))>0.006]
    print(v, d["value"], d["parity_spot_check"], "; ".join(diffs))
PY
