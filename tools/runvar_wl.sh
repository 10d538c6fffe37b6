# usage: WL=hd bash tools/runvar_wl.sh <suffix> ...  -- tuning variants of the library (tools/buildvars.py) on one bench.py workload
cd ${GRAFT_REPO_ROOT:-/root/repo}; WL=${WL:-hd}
for v in "" "$@"; do
  NIK_LIB=$PWD/ni-slam_amd/libnislam_kcc_hip$v.so timeout 300 python bench.py --workload $WL --steps ${STEPS:-20} --warmup 3 --cpu-sample 8 --no-live-prof > gpurun_out/var$v.json 2>gpurun_out/var$v.err || echo "FAIL $v"
done
python - "$@" <<PY
import json,sys
def load(v): return json.loads(open("gpurun_out/var%s.json"%v).read().strip().splitlines()[-1])
base=load(""); bk={k["name"]:k["avg_ms"] for k in base["kernels"]}
print("base", base["value"], base["parity_spot_check"])
for v in sys.argv[1:]:
    try: d=load(v)
    except Exception as e: print(v,"ERR", open("gpurun_out/var%s.err"%v).read()[-300:]); continue
    diffs=["%s %.3f->%.3f"%(k["name"],bk.get(k["name"],0),k["avg_ms"]) for k in d["kernels"] if abs(k["avg_ms"]-bk.get(k["name"],0))>0.008]
    print(v, d["value"], d["parity_spot_check"], "; ".join(diffs))
PY
