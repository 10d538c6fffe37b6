#!/bin/bash
# round-5 evidence run on the GPU box: profiles of the headline and HD workloads (final build), the other workloads, latency,
# parity sweeps of the tiled family at 640x480 and at the two new tiled sizes.  Everything lands under gpurun_out/ (copy the
# files to keep into profiles/).
cd $GRAFT_REPO_ROOT
O=gpurun_out
bash tools/profile_round.sh r05 pairs > $O/r05_profile.log 2>&1
bash tools/profile_round.sh r05_hd hd > $O/r05_hd_profile.log 2>&1
python bench.py --workload sequence --host-frames --cpu-sample 1 > $O/r05_workload_sequence.json 2> $O/r05_workload_sequence.err
python bench.py --workload pyramid > $O/r05_workload_pyramid.json 2> $O/r05_workload_pyramid.err
python bench.py --workload loop4096 > $O/r05_workload_loop4096.json 2> $O/r05_workload_loop4096.err
python tools/latency.py > $O/r05_latency.json 2> $O/r05_latency.err
python tools/parity_sweep.py 1024 10.0 0 > $O/r05_parity_sweep.json 2> $O/r05_parity_sweep.err
python tools/parity_sweep.py 256 10.0 0 480 752 > $O/r05_parity_sweep_752x480_tiled.json 2> $O/r05_parity_sweep_752.err
python tools/parity_sweep.py 256 10.0 0 512 512 > $O/r05_parity_sweep_512x512_tiled.json 2> $O/r05_parity_sweep_512.err
bash tools/pmc_sq.sh > $O/r05_pmc_sq.log 2>&1
python tools/pmc_sq_table.py $O/pmcsq $O/r05_kernel_times.json > $O/r05_pmc_sq_table.csv 2>/dev/null
for f in r05_workload_sequence r05_workload_pyramid r05_workload_loop4096; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['value'], d['path_roofline']['frac_of_8TBps'], d.get('parity_spot_check'), d.get('host_inclusive'))"; done
head -c 600 $O/r05_parity_sweep.json; echo; head -c 600 $O/r05_parity_sweep_752x480_tiled.json; echo; head -c 600 $O/r05_parity_sweep_512x512_tiled.json; echo
python -c "
import json; d=json.load(open('$O/r05_bench.json')); print(d['value'], d['timing'], d['roofline']['frac'], d['roofline'].get('frac_moved_bytes'), d['cpu_baseline'])
d=json.load(open('$O/r05_hd_bench.json')); print('hd', d['value'], d['path_roofline'])"
