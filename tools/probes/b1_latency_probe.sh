# B = 1 latency by measurement: the rotation stage's four phases as four dependent launches against ONE
# persistent launch whose phases hand over through per-item counters (tools/probes/xcd_pipeline_probe.hip, the synthetic stage
# that reproduces the real kernels' time to 4 % at 256 items), at 1 / 2 / 8 items per batch.
cd ${GRAFT_REPO_ROOT:-/root/repo}
BIN=tools/probes/xcd_pipeline_probe
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN tools/probes/xcd_pipeline_probe.hip || exit 1
for n in 1 2 8; do for wpc in 5 2; do
  echo "== items $n, $wpc workgroups per CU"
  timeout 120 $BIN --items $n --reps 300 --mode 5 --D 1 --G 1 --wpc $wpc | grep -v "^rotation-stage"
done; done
