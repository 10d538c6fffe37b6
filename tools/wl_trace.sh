# usage: WL=pyramid bash tools/wl_trace.sh  -- rocprofv3 kernel trace of a bench.py workload, summarised by tools/seq_trace.py
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
WL=${WL:-pyramid}; mkdir -p gpurun_out/trace_$WL
timeout 500 rocprofv3 --kernel-trace -d gpurun_out/trace_$WL -o t -- python bench.py --workload $WL --steps ${STEPS:-40} --warmup 5 --repeats 1 --cpu-sample 0 --no-profile --no-live-prof > gpurun_out/trace_$WL/bench.log 2>&1 || echo "rc=$?"
grep '^{' gpurun_out/trace_$WL/bench.log | cut -c1-260
GAP=${GAP:-1.5e6} python tools/seq_trace.py gpurun_out/trace_$WL/t_results.db | tee gpurun_out/trace_$WL/summary.txt
rm -f gpurun_out/trace_$WL/t_results.db
