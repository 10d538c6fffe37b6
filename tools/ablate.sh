# usage: bash tools/ablate.sh  -- per-kernel time with loads / stores / FFT removed (debug ablation flags)
cd $GRAFT_REPO_ROOT
for a in 0 3 4 7 15; do
  NIK_ABLATE=$a python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-cached > gpurun_out/abl$a.json 2>gpurun_out/abl$a.err || echo "FAIL $a"
done
python - <<PY
import json
d={a:{k["name"]:k["avg_ms"] for k in json.load(open("gpurun_out/abl%d.json"%a))["kernels"]} for a in (0,3,4,7,15)}
print("%-28s %7s %7s %7s %7s %7s"%("kernel","full","noLDST","noFFT","none","exit"))
for k in d[0]: print("%-28s "%k+" ".join("%7.3f"%d[a].get(k,0) for a in (0,3,4,7,15)))
PY
