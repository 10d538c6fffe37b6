#!/bin/bash
# round-4 evidence run on the GPU box: profiles of the headline and HD workloads, the other workloads, latency, parity sweeps of
# the tiled and the any-size family.  Everything lands under gpurun_out/ (copy the files to keep into profiles/).
cd $GRAFT_REPO_ROOT
O=gpurun_out
bash tools/profile_round.sh r04 pairs > $O/r04_profile.log 2>&1
bash tools/profile_round.sh r04_hd hd > $O/r04_hd_profile.log 2>&1
python bench.py --workload sequence --host-frames --cpu-sample 1 > $O/r04_workload_sequence.json 2> $O/r04_workload_sequence.err
python bench.py --workload sequence --seq-motion smooth --cpu-sample 0 > $O/r04_workload_sequence_smooth.json 2>/dev/null
python bench.py --workload pyramid > $O/r04_workload_pyramid.json 2> $O/r04_workload_pyramid.err
python bench.py --workload loop4096 > $O/r04_workload_loop4096.json 2> $O/r04_workload_loop4096.err
python bench.py --workload hd --no-live-prof > $O/r04_workload_hd.json 2> $O/r04_workload_hd.err
python tools/latency.py > $O/r04_latency.json 2> $O/r04_latency.err
python tools/parity_sweep.py 1024 10.0 0 > $O/r04_parity_sweep.json 2> $O/r04_parity_sweep.err
python tools/parity_sweep.py 256 10.0 0 480 752 > $O/r04_parity_sweep_752x480_generic.json 2> $O/r04_parity_sweep_752.err
NIK_GENERIC=1 python tools/parity_sweep.py 512 10.0 0 > $O/r04_parity_sweep_generic_640x480.json 2> $O/r04_parity_sweep_generic.err
NIK_GENERIC=1 python bench.py --no-live-prof --no-profile --cpu-sample 16 --no-cached --repeats 3 > $O/r04_generic_family_bench.json 2>/dev/null
bash tools/pmc_sq.sh > $O/r04_pmc_sq.log 2>&1
python tools/pmc_sq_table.py $O/pmcsq $O/r04_kernel_times.json > $O/r04_pmc_sq_table.csv 2>/dev/null
for f in r04_workload_sequence r04_workload_pyramid r04_workload_loop4096 r04_workload_hd r04_generic_family_bench; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['value'], d['path_roofline']['frac_of_8TBps'], d.get('parity_spot_check'), d.get('host_inclusive'))"; done
head -c 600 $O/r04_parity_sweep.json; echo; head -c 600 $O/r04_parity_sweep_752x480_generic.json; echo; head -c 400 $O/r04_parity_sweep_generic_640x480.json; echo
python -c "
import json; d=json.load(open('$O/r04_bench.json')); print(d['value'], d['timing'], d['roofline']['frac'], d['roofline'].get('frac_moved_bytes'))"
