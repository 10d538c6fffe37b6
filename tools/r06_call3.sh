# round 6, GPU call 3: HD compile-time variants of the 1280-point B kernels; upper bounds by deletion on the headline kernels;
# u8 ablation bits; sequence window sweep; the new default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
B="python bench.py --cpu-sample 0 --no-live-prof --no-cached --steps 20 --warmup 3"
L=$PWD/ni-slam_amd/libnislam_kcc_hip
kern() { python - "$@" <<PY
import json,sys
base=json.load(open(sys.argv[1])); bk={k["name"]:k["avg_ms"] for k in base["kernels"]}
print("base %s: %.1f"%(sys.argv[1].split("/")[-1], base["value"]))
for f in sys.argv[2:]:
    try: d=json.load(open(f))
    except Exception as e: print(f.split("/")[-1],"ERR",e); continue
    print("%-22s %9.1f  "%(f.split("/")[-1], d["value"]) + "; ".join("%s %.4f->%.4f"%(k["name"],bk.get(k["name"],0),k["avg_ms"]) for k in d["kernels"] if abs(k["avg_ms"]-bk.get(k["name"],0))>0.004))
PY
}
# 1. HD variants
for v in "" _seq1280 _huge2x2 _huge1 _huge4; do NIK_LIB=$L$v.so $B --workload hd > $O/hd$v.json 2> $O/hd$v.err || echo "FAIL hd $v"; done
kern $O/hd.json $O/hd_seq1280.json $O/hd_huge2x2.json $O/hd_huge1.json $O/hd_huge4.json | tee $O/hd_variants.txt
# 2. upper bounds by deletion (tuning builds; batch 256, 3 streams: the round-5 reference point)
for v in _tune _ubmom _ubzz _ubpol; do NIK_LIB=$L$v.so $B --batch 256 --streams 3 > $O/ub$v.json 2> $O/ub$v.err || echo "FAIL ub $v"; done
kern $O/ub_tune.json $O/ub_ubmom.json $O/ub_ubzz.json $O/ub_ubpol.json | tee $O/upper_bounds.txt
# 3. u8 kernel ablation: 1 no loads, 64 no frame-store copy, 65 both
for a in 0 1 64 65 2; do NIK_LIB=${L}_tune.so NIK_ABLATE=$a $B --batch 256 --streams 3 --repeats 1 > $O/u8abl$a.json 2> $O/u8abl.err || echo "FAIL u8abl $a"; done
python - <<PY | tee $O/u8_ablate.txt
import json
print("# kA_fwd<240,u8> / rot8 / polar, ms per 256 pairs; NIK_ABLATE 0 full, 1 no loads/gathers, 64 no u8 frame-store copy, 65 both, 2 no stores")
for a in (0,1,64,65,2):
    try:
        k={x["name"]:x["avg_ms"] for x in json.load(open("$O/u8abl%d.json"%a))["kernels"]}
        print("%3d  u8 %.4f  rot8 %.4f  polar %.4f  argmax240 %.4f argmax360 %.4f"%(a,k["kA_fwd<240,u8>"],k["kA_fwd<240,rot8>"],k["kA_fwd<360,polar>"],k["kA_inv<240,argmax>"],k["kA_inv<360,argmax>"]))
    except Exception as e: print(a,"ERR",e)
PY
# 4. sequence: window x look-ahead depth (tuning library for the depth switch)
for w in 64 128 256; do for d in 2 3; do
  NIK_LIB=${L}_tune.so NIK_SEQ_WINDOW=$w NIK_TRK_DEPTH=$d timeout 300 python bench.py --workload sequence --batch $w --steps 10 --cpu-sample 0 > $O/seq_w${w}_d$d.json 2> $O/seq.err || echo "FAIL seq $w $d"
  python - $O/seq_w${w}_d$d.json $w $d <<PY | tee -a $O/seq_sweep.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("window %s depth %s: %.0f frames/s  calls %d  enqueued %d consumed %d"%(sys.argv[2],sys.argv[3],d["value"],c["batched_pose_calls"],c["registrations"]["pairs_enqueued"],c["registrations"]["pairs_consumed"]))
except Exception as e: print("window",sys.argv[2],"depth",sys.argv[3],"FAILED",e)
PY
done; done
# 5. the default line of this build (batch 512, 2 streams)
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print(d["value"], d["path_roofline"]["frac_of_8TBps"], d["config"]["pairs_per_gpu_per_step"], d["config"]["streams_per_gpu"], d["parity_spot_check"], d["cpu_baseline"])
print(json.dumps(d["roofline"]))
for k in d["kernels"]: print("  %-28s %.4f ms  design %.0f MB  %6.0f GB/s (nominal %6.0f)"%(k["name"],k["avg_ms"],k["design_bytes_per_launch"]/1e6,k["gbps"],k["gbps_nominal"]))
PY
