# chunked calls at the new headline batch: do intermediates that fit the 256 MiB Infinity Cache pay now? (round 3 measured: no, at 256 pairs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
B="python bench.py --cpu-sample 0 --no-live-prof --no-cached --no-profile --steps 20 --warmup 3 --unique 256"
for cfg in "512 2 0" "512 2 128" "512 2 64" "512 3 64" "512 4 64" "512 4 32" "1024 2 128" "1024 4 64" "512 2 0"; do set -- $cfg
  $B --batch $1 --streams $2 --chunk $3 > $O/c.json 2> $O/c.err || echo FAIL $cfg
  python -c "
import json; d=json.load(open('$O/c.json')); t=d['timing']; print('batch %4s streams %s chunk %3s: %9.1f [%9.1f, %9.1f]'%('$1','$2','$3',d['value'],t['value_min'],t['value_max']))"
done 2>&1 | tee $O/chunk_sweep.txt
