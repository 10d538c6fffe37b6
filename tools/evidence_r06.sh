#!/bin/bash
# round-6 evidence run on the GPU box (final build): profiles of the headline (512 pairs x 2 streams) and HD workloads, the other
# workloads, latency, parity sweeps with the theta letter counts, SQ counter table, the first_8gpu script's self test.
# Everything lands under gpurun_out/ (copy the files to keep into profiles/).
cd $GRAFT_REPO_ROOT
O=gpurun_out
bash tools/profile_round.sh r06 pairs > $O/r06_profile.log 2>&1
bash tools/profile_round.sh r06_hd hd > $O/r06_hd_profile.log 2>&1
python bench.py --workload sequence --host-frames --cpu-sample 1 > $O/r06_workload_sequence.json 2> $O/r06_workload_sequence.err
python bench.py --workload pyramid > $O/r06_workload_pyramid.json 2> $O/r06_workload_pyramid.err
python bench.py --workload loop4096 > $O/r06_workload_loop4096.json 2> $O/r06_workload_loop4096.err
python tools/latency.py > $O/r06_latency.json 2> $O/r06_latency.err
python tools/parity_sweep.py 1024 10.0 0 > $O/r06_parity_sweep.json 2> $O/r06_parity_sweep.err
python tools/parity_sweep.py 128 8.0 0 720 1280 > $O/r06_parity_sweep_hd.json 2> $O/r06_parity_sweep_hd.err
bash tools/pmc_sq.sh > $O/r06_pmc_sq.log 2>&1
python tools/pmc_sq_table.py $O/pmcsq $O/r06_kernel_times.json > $O/r06_pmc_sq_table.csv 2>/dev/null
python tools/first_8gpu.py --selftest --out $O/r06_first_8gpu_selftest.json > $O/r06_first_8gpu_selftest.txt 2>&1; tail -12 $O/r06_first_8gpu_selftest.txt
python tools/soak.py > $O/r06_soak.txt 2>&1; tail -3 $O/r06_soak.txt
for f in r06_workload_sequence r06_workload_pyramid r06_workload_loop4096; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['value'], d['path_roofline']['frac_of_8TBps'], d.get('parity_spot_check'), d.get('host_inclusive'))"; done
python - <<PY
import json
for f in ("r06_parity_sweep","r06_parity_sweep_hd"):
    d=json.load(open("$O/%s.json"%f))
    for m in ("small_rot","large_rot"):
        r=d[m]; print(f, m, {k:r[k] for k in r if k not in ("translation_near_ties(gap,pixels)","first_failures","theta_difference_examples")})
d=json.load(open("$O/r06_bench.json")); print(d["value"], d["timing"], json.dumps(d["roofline"]), d["parity_spot_check"], d["cpu_baseline"], d.get("kzz_cached_mode"))
for k in d["kernels"]: print("  %-28s %.4f ms  design %.0f MB  %6.0f GB/s (nominal %6.0f)"%(k["name"],k["avg_ms"],k["design_bytes_per_launch"]/1e6,k["gbps"],k["gbps_nominal"]))
d=json.load(open("$O/r06_hd_bench.json")); print("hd", d["value"], d["path_roofline"], d["parity_spot_check"])
for k in d["kernels"]: print("  %-32s %.4f ms  %6.0f GB/s"%(k["name"],k["avg_ms"],k["gbps"]))
PY
cat $O/r06_pmc_sq_table.csv
