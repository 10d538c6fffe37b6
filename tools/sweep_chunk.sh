# usage (GPU box): bash tools/sweep_chunk.sh <outdir>   -- pairs/s of the default workload against chunk size, item order and streams;
# every variant's per-pair results of the last step must equal the first variant's (outputs do not depend on the scheduling)
OUT=${1:-gpurun_out/sweep}
mkdir -p $OUT
for S in ${STREAMS:-2 3}; do for A in 0 1; do for C in ${CHUNKS:-0 24 32 48 64 96 128}; do
  T=s${S}_a${A}_c${C}
  NIK_BENCH_DUMP=$OUT/$T.dump NIK_STREAMS=$S NIK_ALT_ORDER=$A NIK_CHUNK=$C python bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-profile --no-cached > $OUT/$T.json 2> $OUT/$T.err
  python - <<PY
import json, glob
try:
    d = json.load(open("$OUT/$T.json"))
    ref = sorted(glob.glob("$OUT/*.dump.0"))[0]
    same = json.load(open(ref))["results"] == json.load(open("$OUT/$T.dump.0"))["results"]
    print("streams $S alt $A chunk $C:", round(d["value"]), "same results as", ref.split("/")[-1], same)
except Exception as e:
    print("streams $S alt $A chunk $C: FAILED", e)
PY
done; done; done
