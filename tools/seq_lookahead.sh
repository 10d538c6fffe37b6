# usage: bash tools/seq_lookahead.sh  -- the tracker's look-ahead batches on the sequence workload: depth x batch-room sweep
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/seqla
timeout 900 python -m pytest tests/test_tracker.py -m gpu -x -q > gpurun_out/seqla/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/seqla/pytest.log
for D in 1 2 3 4; do for F in 0 32; do
  NIK_TRK_DEPTH=$D NIK_TRK_FLIGHT=$F timeout 300 python bench.py --workload sequence --steps 20 --cpu-sample 0 > gpurun_out/seqla/seq_d${D}_f${F}.json 2> gpurun_out/seqla/seq_d${D}_f${F}.err
  python - gpurun_out/seqla/seq_d${D}_f${F}.json $D $F <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("depth %s room %s: %.0f frames/s  held %d failed %d calls %d  graph %s"%(sys.argv[2],sys.argv[3],d["value"],c["keyframe_guesses_held"],c["keyframe_guesses_failed"],c["batched_pose_calls"],d["hipgraph"]))
except Exception as e: print("depth %s room %s: FAILED %s"%(sys.argv[2],sys.argv[3],e))
PY
done; done
NIK_SEQ_WINDOW=128 timeout 300 python bench.py --workload sequence --steps 20 --cpu-sample 0 --batch 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('window 128:', d['value'], d['config'])"
timeout 600 python bench.py --workload sequence --steps 20 --host-frames > gpurun_out/seqla/seq_default.json 2> gpurun_out/seqla/seq_default.err; tail -c 1500 gpurun_out/seqla/seq_default.json
