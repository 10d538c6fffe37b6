# usage: bash tools/seq_trace.sh  -- rocprofv3 kernel trace of the sequence workload (sqlite; analysed by tools/seq_trace.py)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/seqtrace
timeout 500 rocprofv3 --kernel-trace -d gpurun_out/seqtrace -o seq -- python bench.py --workload sequence --steps 4 --warmup 2 --repeats 1 --cpu-sample 0 > gpurun_out/seqtrace/bench.log 2>&1 || echo "rc=$?"
grep '^{' gpurun_out/seqtrace/bench.log | cut -c1-200
python tools/seq_trace.py gpurun_out/seqtrace/seq_results.db | tee gpurun_out/seqtrace/summary.txt
