# usage: bash tools/ablate.sh [bits ...]  -- per-kernel time with parts of the kernels removed (ablation build of the library:
# tools/buildvars.py "abl=-DKCC_ABLATE"; bits: kcc_kernels.hip "Performance ablation")
cd $GRAFT_REPO_ROOT
BITS="${@:-0 3 4 7 15}"
for a in $BITS; do
  NIK_LIB=$PWD/ni-slam_amd/libnislam_kcc_hip_tune.so NIK_ABLATE=$a python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-cached > gpurun_out/abl$a.json 2>gpurun_out/abl$a.err || echo "FAIL $a"
done
python - $BITS <<PY
import json, sys
bits=[int(b) for b in sys.argv[1:]]
d={a:{k["name"]:k["avg_ms"] for k in json.load(open("gpurun_out/abl%d.json"%a))["kernels"]} for a in bits}
print("%-28s "%"kernel"+" ".join("%7d"%a for a in bits))
for k in d[bits[0]]: print("%-28s "%k+" ".join("%7.3f"%d[a].get(k,0) for a in bits))
PY
