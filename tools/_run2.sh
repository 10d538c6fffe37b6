cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/seqla2
for M in sawtooth smooth; do for D in 1 2; do
NIK_TRK_DEPTH=$D timeout 300 python bench.py --workload sequence --steps 20 --cpu-sample 0 --seq-motion $M 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$M depth $D: %.0f frames/s held %d failed %d calls %d keyframes %d'%(d['value'],c['keyframe_guesses_held'],c['keyframe_guesses_failed'],c['batched_pose_calls'],c['keyframes']))"
done; done
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/seqla2/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/seqla2/pytest_gpu.log
