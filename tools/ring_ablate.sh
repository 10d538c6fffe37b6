# usage: bash tools/ring_ablate.sh  -- what the ring-form B kernels' time is made of, next to the one-tile-per-workgroup kernels:
# ablation build (tools/buildvars.py "abl=-DKCC_ABLATE"), $NIK_ABLATE 0 full / 4 no FFT (data movement only) / 3 no loads, no
# stores (arithmetic + LDS only) / 1 no loads / 2 no stores, for $NIK_RING 0 and 7
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for r in 0 7; do for a in 0 4 3 1 2; do
  NIK_LIB=$PWD/ni-slam_amd/libnislam_kcc_hip_tune.so NIK_RING=$r NIK_ABLATE=$a timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-cached --no-live-prof > gpurun_out/rabl_${r}_$a.json 2>gpurun_out/rabl_${r}_$a.err || echo "FAIL $r $a"
done; done
python - <<PY
import json
def load(r,a):
    try: return {k["name"]:k["avg_ms"] for k in json.loads(open("gpurun_out/rabl_%d_%d.json"%(r,a)).read().strip().splitlines()[-1])["kernels"]}
    except Exception as e: return {}
bits=[0,4,3,1,2]; lab={0:"full",4:"noFFT",3:"noLdSt",1:"noLd",2:"noSt"}
print("%-26s "%"kernel (ms / 256 pairs)"+" ".join("%8s"%("r0 "+lab[a]) for a in bits)+" | "+" ".join("%8s"%("r7 "+lab[a]) for a in bits))
d={(r,a):load(r,a) for r in (0,7) for a in bits}
for k in d[(0,0)]:
    if k.startswith("kB<"):
        print("%-26s "%k+" ".join("%8.4f"%d[(0,a)].get(k,0) for a in bits)+" | "+" ".join("%8.4f"%d[(7,a)].get(k,0) for a in bits))
PY
