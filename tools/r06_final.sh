# final pass of round 6: headline profile + parity sweeps (theta letter at float precision) + the -m gpu suite of the final tree
cd $GRAFT_REPO_ROOT
O=gpurun_out
bash tools/profile_round.sh r06 pairs > $O/r06_profile.log 2>&1
python tools/parity_sweep.py 1024 10.0 0 > $O/r06_parity_sweep.json 2> $O/r06_parity_sweep.err
python tools/parity_sweep.py 128 8.0 0 720 1280 > $O/r06_parity_sweep_hd.json 2> $O/r06_parity_sweep_hd.err
bash tools/pmc_sq.sh > $O/r06_pmc_sq.log 2>&1
python tools/pmc_sq_table.py $O/pmcsq $O/r06_kernel_times.json > $O/r06_pmc_sq_table.csv 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q > $O/r06_gpu_suite.log 2>&1; tail -3 $O/r06_gpu_suite.log
python -c "
import json
d=json.load(open('$O/r06_bench.json')); print(d['value'], d['timing']['value_min'], d['timing']['value_max'], d['roofline']['frac'], d['parity_spot_check'])
for f in ('r06_parity_sweep','r06_parity_sweep_hd'):
    s=json.load(open('$O/%s.json'%f))
    for m in ('small_rot','large_rot'): r=s[m]; print(f,m,r['exact'],r['mirror_tie_accepted'],r['other_near_tie_verified'],r['failed'],r['theta_equal_to_oracle'],r['theta_differs_by_2pi'],r['theta_differs_with_identical_rotation_rows'],r['worst_psr_rel_err'])
"
