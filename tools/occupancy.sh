# usage (GPU box): bash tools/occupancy.sh  -- kernel time against workgroups per CU: the ablation build pads the dynamic LDS
# of the B kernels (NIK_LDS_PAD_B) or the inverse A kernels (NIK_LDS_PAD_A) so that fewer workgroups fit on a CU
cd $GRAFT_REPO_ROOT
for v in "B 0" "B 6000" "B 18000" "B 28000" "A 0" "A 3000" "A 16000" "A 26000"; do
  set -- $v
  env NIK_LDS_PAD_$1=$2 NIK_LIB=$PWD/ni-slam_amd/libnislam_kcc_hip_tune.so NIK_ABLATE=0 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-cached > gpurun_out/occ_$1_$2.json 2>gpurun_out/occ.err || echo FAIL $v
done
python - <<PY
import json
for fam, pads in (("B", (0, 6000, 18000, 28000)), ("A", (0, 3000, 16000, 26000))):
    d = {p: {k["name"]: k["avg_ms"] for k in json.load(open("gpurun_out/occ_%s_%d.json" % (fam, p)))["kernels"]} for p in pads}
    print("%-28s " % ("LDS pad " + fam) + " ".join("%7d" % p for p in pads))
    for k in d[0]:
        if k.startswith("k" + fam) and (fam == "B" or "kA_inv" in k): print("%-28s " % k + " ".join("%7.3f" % d[p].get(k, 0) for p in pads))
PY
