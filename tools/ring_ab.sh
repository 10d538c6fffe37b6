# usage: bash tools/ring_ab.sh [mask ...]  -- the ring-form B kernels ($NIK_RING bit mask: 1 fwd_abs_inv, 2 (fwd_)mul_inv, 4 solve_inv)
# against the one-tile-per-workgroup kernels on the headline workload: pairs/s and the per-kernel HIP-event times
cd ${GRAFT_REPO_ROOT:-/root/repo}; WL=${WL:-pairs}; mkdir -p gpurun_out
MASKS="${@:-0 1 2 4 7}"
# an entry is mask or mask_suffix (tuning variant of the library, tools/buildvars.py)
for m in $MASKS; do
  v=${m#*_}; [ "$v" = "$m" ] && v="" || v="_$v"
  NIK_LIB=$PWD/ni-slam_amd/libnislam_kcc_hip$v.so NIK_RING=${m%%_*} timeout 300 python bench.py --workload $WL --steps ${STEPS:-20} --warmup 3 --cpu-sample 8 --no-live-prof --no-cached > gpurun_out/ring$m.json 2>gpurun_out/ring$m.err || echo "FAIL $m"
done
python - $MASKS <<PY
import json,sys
def load(v): return json.loads(open("gpurun_out/ring%s.json"%v).read().strip().splitlines()[-1])
ms=sys.argv[1:]; d={}
for m in ms:
    try: d[m]=load(m)
    except Exception as e: print(m,"ERR",open("gpurun_out/ring%s.err"%m).read()[-400:])
ms=[m for m in ms if m in d]
print("%-30s "%"NIK_RING"+" ".join("%9s"%m for m in ms))
print("%-30s "%"pairs/s"+" ".join("%9.0f"%d[m]["value"] for m in ms))
print("%-30s "%"parity"+" ".join("%9s"%str(d[m].get("parity_spot_check")) for m in ms))
names=[k["name"] for k in d[ms[0]]["kernels"]]
for n in names:
    print("%-30s "%n+" ".join("%9.4f"%next((k["avg_ms"] for k in d[m]["kernels"] if k["name"]==n),0) for m in ms))
print("%-30s "%"sum"+" ".join("%9.4f"%sum(k["avg_ms"] for k in d[m]["kernels"]) for m in ms))
PY
