cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
export NIK_LIB=$PWD/ni-slam_amd/libnislam_kcc_hip_tune.so
B="python bench.py --cpu-sample 16 --no-live-prof --no-cached --steps 20 --warmup 3 --batch 256 --streams 3"
for m in 0 1 4 8 16 0 4 8; do NIK_POLAR_ALIGNED=$m $B > $O/pal_$m.json 2> $O/pal.err || echo FAIL $m; python - $O/pal_$m.json $m <<PY
import json,sys
d=json.load(open(sys.argv[1])); k={x["name"]:x["avg_ms"] for x in d["kernels"]}
print("NIK_POLAR_ALIGNED=%s: %.1f pairs/s  polar %.4f ms  parity %s"%(sys.argv[2], d["value"], k["kA_fwd<360,polar>"], d["parity_spot_check"]["ok"]))
PY
done 2>&1 | tee $O/polar_aligned.txt
NIK_POLAR_ALIGNED=1 NIK_UNDER_TUNING_LIB=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "polar or gather or pose_parity" 2>&1 | tail -3
