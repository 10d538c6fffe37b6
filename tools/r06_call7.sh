cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
L=$PWD/ni-slam_amd/libnislam_kcc_hip
for rep in 1 2 3; do for v in "" _trkold; do
  NIK_LIB=$L$v.so python bench.py --workload sequence --host-frames --cpu-sample 0 --steps 10 > $O/s$v.$rep.json 2> $O/s.err || echo FAIL $v
  python - $O/s$v.$rep.json "${v:-new(window 0 prefetched)}" <<PY
import json,sys
d=json.load(open(sys.argv[1])); h=d["host_inclusive"]
print("%-28s resident %.0f  pinned %.0f  pageable %.0f  identical %s"%(sys.argv[2], h["frames_per_s_resident"], h["frames_per_s_pinned_source"], h["frames_per_s_pageable_source"], h["identical_outputs"]))
PY
done; done 2>&1 | tee $O/push_host_marker.txt
( time python bench.py > $O/default_bench.json 2> $O/default_bench.err ) 2>&1 | grep real
