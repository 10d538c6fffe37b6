cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
L=$PWD/ni-slam_amd/libnislam_kcc_hip
B="python bench.py --cpu-sample 0 --no-live-prof --no-cached --steps 20 --warmup 3"
for rep in 1 2 3; do for v in "" _pal0 _pal8; do
  NIK_LIB=$L$v.so $B > $O/p$v.$rep.json 2> $O/p.err || echo FAIL $v
  python - $O/p$v.$rep.json "${v:-_pal16(release)}" <<PY
import json,sys
d=json.load(open(sys.argv[1])); k={x["name"]:x["avg_ms"] for x in d["kernels"]}
print("%-18s %.1f pairs/s [%.1f, %.1f]  polar %.4f ms per 512"%(sys.argv[2], d["value"], d["timing"]["value_min"], d["timing"]["value_max"], k["kA_fwd<360,polar>"]))
PY
done; done 2>&1 | tee $O/polar_aligned_release.txt
