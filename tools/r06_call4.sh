# round 6, GPU call 4: upper bounds on the RELEASE code generation; gather ablation split (staging / sampling); theta letter in a parity sweep; GPU suite of the current tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
B="python bench.py --cpu-sample 0 --no-live-prof --no-cached --steps 20 --warmup 3 --batch 256 --streams 3"
L=$PWD/ni-slam_amd/libnislam_kcc_hip
for v in "" _ubmomr _ubzzr _ubpolr ""; do NIK_LIB=$L$v.so $B > $O/ubr$v.json 2> $O/ubr$v.err || echo "FAIL ubr $v"; [ -z "$v" ] && cp $O/ubr.json $O/ubr_base_$(date +%s).json; done
python - $O/ubr.json $O/ubr_ubmomr.json $O/ubr_ubzzr.json $O/ubr_ubpolr.json <<PY | tee $O/upper_bounds_release.txt
import json,sys,glob
print("# upper bounds by deletion on the release code generation (-DKCC_UB_* -DKCC_UB_TIMING_ONLY), batch 256, 3 streams; pairs/s; kernels that moved > 3 us (ms per 256 pairs)")
for f in sorted(glob.glob("$O/ubr_base_*.json")): print("base run", json.load(open(f))["value"])
base=json.load(open(sys.argv[1])); bk={k["name"]:k["avg_ms"] for k in base["kernels"]}
for f in sys.argv[2:]:
    d=json.load(open(f)); print("%-18s %9.1f  "%(f.split("/")[-1], d["value"]) + "; ".join("%s %.4f->%.4f"%(k["name"],bk.get(k["name"],0),k["avg_ms"]) for k in d["kernels"] if abs(k["avg_ms"]-bk.get(k["name"],0))>0.003))
PY
for a in 0 16 32 48 1; do NIK_LIB=${L}_tune.so NIK_ABLATE=$a $B --repeats 1 > $O/gabl$a.json 2> $O/gabl.err || echo "FAIL gabl $a"; done
python - <<PY | tee $O/gather_split.txt
import json
print("# gathers, tuning build, ms per 256 pairs; NIK_ABLATE 0 full, 16 no staging (LDS-DMA), 32 no sampling, 48 neither, 1 no gather at all")
for a in (0,16,32,48,1):
    try:
        k={x["name"]:x["avg_ms"] for x in json.load(open("$O/gabl%d.json"%a))["kernels"]}
        print("%3d  rot8 %.4f  polar %.4f  u8 %.4f"%(a,k["kA_fwd<240,rot8>"],k["kA_fwd<360,polar>"],k["kA_fwd<240,u8>"]))
    except Exception as e: print(a,"ERR",e)
PY
python tools/parity_sweep.py 256 10.0 0 > $O/parity_sweep_256.json 2> $O/parity_sweep.err; python - <<PY
import json; d=json.load(open("$O/parity_sweep_256.json"))
for m in ("small_rot","large_rot"):
    r=d[m]; print(m, {k:r[k] for k in r if k not in ("translation_near_ties(gap,pixels)","first_failures")})
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; tail -5 $O/gpu_suite.log
