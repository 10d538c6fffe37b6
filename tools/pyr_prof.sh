# usage (GPU box): bash tools/pyr_prof.sh  -- kernel-time sum per pyramid step vs wall time
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pyr -- python $R/bench.py --workload ${WL:-pyramid} --steps 50 --warmup 5 --cpu-sample 0 > $R/gpurun_out/pyr.log 2>&1
grep '"metric"' $R/gpurun_out/pyr.log | cut -c1-260
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/pyr/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("total kernel ms", tot/1e6, "calls", calls, "per step (55 steps): ms", tot/1e6/55, "launches", calls/55)
for r in rows[:30]: print(r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3)
PY
python - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/pyr/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Queue_Id"] if "Queue_Id" in r else r.get("Stream_Id","?")) for r in rows)
t0,t1=ev[len(ev)//5][0],ev[-len(ev)//10][0]
ev=[e for e in ev if t0<=e[0]<t1]
busy=0;cur_s,cur_e=ev[0][0],ev[0][1]
for s,e,_ in ev[1:]:
    if s>cur_e: busy+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print("window ms", (t1-t0)/1e6, "GPU busy (union) frac", busy/(t1-t0), "sum of kernel time / window", sum(e-s for s,e,_ in ev)/(t1-t0))
q=collections.defaultdict(int)
for s,e,k in ev: q[k]+=e-s
print({k:round(v/(t1-t0),3) for k,v in q.items()})
PY
