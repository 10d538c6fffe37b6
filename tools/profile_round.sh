# usage (on the GPU box, via gpurun):  bash tools/profile_round.sh r03a [workload]
# Produces the round's evidence under gpurun_out/ AND profiles/ (tools/rocprof_summary.py: three separate rocprofv3 passes):
#   <tag>_kernel_stats.csv       rocprofv3 --kernel-trace --stats of the bench command (1 stream, uncached path)
#   <tag>_kernel_times.json      the same per stage name (what bench.py's roofline cross-checks its HIP-event durations with)
#   <tag>_pmc_hbm_summary.csv, <tag>_pmc_traffic.json   FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs)
#   <tag>_bench.json             the default bench line of the same build
TAG=${1:-rXX}
WL=${2:-pairs}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
B=$([ "$WL" = hd ] && echo 128 || echo 512)
python $R/tools/rocprof_summary.py $TAG $OUT $B -- python $R/bench.py --workload $WL --steps 120 --warmup 40 --cpu-sample 0 --no-profile --no-cached --no-live-prof
# (gpurun merges gpurun_out/ back, not profiles/: copy gpurun_out/${TAG}_* into profiles/ afterwards)
cd $R && python bench.py --workload $WL --steps 30 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python -c "
import json; d=json.load(open('$OUT/${TAG}_bench.json')); print(d['value'], d['roofline'], d['cpu_baseline'], d.get('kzz_cached_mode'))"
head -20 $OUT/${TAG}_kernel_stats.csv
