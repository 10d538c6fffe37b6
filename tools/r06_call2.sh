# round 6, GPU call 2: headline batch x streams sweep; HD ablation table; HD compile-time variants; sequence kernel trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
B="python bench.py --cpu-sample 0 --no-live-prof --no-cached --steps 20 --warmup 3"
echo "# headline: pairs/s (median [min,max]) by pairs per step and streams" > $O/batch_sweep.txt
for b in 256 384 512 768 1024; do for s in 1 2 3; do
  NIK_STREAMS=$s $B --no-profile --batch $b --unique 256 > $O/bs_${b}_$s.json 2> $O/bs.err || echo "FAIL $b $s" >> $O/batch_sweep.txt
  python - $O/bs_${b}_$s.json $b $s >> $O/batch_sweep.txt <<PY
import json,sys
try:
    j=json.load(open(sys.argv[1])); t=j["timing"]; print("batch %4s streams %s: %9.1f [%9.1f, %9.1f]"%(sys.argv[2],sys.argv[3],j["value"],t["value_min"],t["value_max"]))
except Exception as e: print("batch",sys.argv[2],"streams",sys.argv[3],"ERR",e)
PY
done; done
cat $O/batch_sweep.txt
# HD ablation (tuning library): 0 full, 1 no loads/gathers, 2 no stores, 3 neither, 4 no FFT, 7 nothing
for a in 0 1 2 3 4 7; do
  NIK_LIB=$PWD/ni-slam_amd/libnislam_kcc_hip_tune.so NIK_ABLATE=$a $B --workload hd --repeats 1 > $O/hd_abl$a.json 2> $O/hd_abl.err || echo "FAIL hd abl $a"
done
python - > $O/hd_ablate.txt <<PY
import json
bits=[0,1,2,3,4,7]
d={}
for a in bits:
    try: d[a]={k["name"]:k["avg_ms"] for k in json.load(open("$O/hd_abl%d.json"%a))["kernels"]}
    except Exception as e: d[a]={}
print("# HD (1280x720, 128 pairs): ms per launch; NIK_ABLATE 0 full, 1 no loads, 2 no stores, 3 neither, 4 no FFT, 7 nothing")
print("%-32s "%"kernel"+" ".join("%7d"%a for a in bits))
for k in d[0]: print("%-32s "%k+" ".join("%7.3f"%d[a].get(k,0) for a in bits))
PY
cat $O/hd_ablate.txt
# HD variants
for v in "" _flx8 _rotwps2 _u8tpw1 _u8tpw4; do
  NIK_LIB=$PWD/ni-slam_amd/libnislam_kcc_hip$v.so $B --workload hd > $O/hdvar$v.json 2> $O/hdvar$v.err || echo "FAIL hdvar $v"
done
python - > $O/hd_variants.txt <<PY
import json
base=json.load(open("$O/hdvar.json")); bk={k["name"]:k["avg_ms"] for k in base["kernels"]}
print("base", base["value"])
for v in ["_flx8","_rotwps2","_u8tpw1","_u8tpw4"]:
    try: d=json.load(open("$O/hdvar%s.json"%v))
    except Exception as e: print(v,"ERR",open("$O/hdvar%s.err"%v).read()[-300:]); continue
    print(v, d["value"], "; ".join("%s %.3f->%.3f"%(k["name"],bk.get(k["name"],0),k["avg_ms"]) for k in d["kernels"] if abs(k["avg_ms"]-bk.get(k["name"],0))>0.006))
PY
cat $O/hd_variants.txt
bash tools/seq_trace.sh > $O/seq_trace.txt 2>&1; cp gpurun_out/seqtrace/summary.txt $O/seq_trace_summary.txt 2>/dev/null; tail -40 $O/seq_trace.txt
