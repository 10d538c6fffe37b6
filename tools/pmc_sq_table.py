"""Per-stage table of the SQ counter passes of tools/pmc_sq.sh (gpurun_out/pmcsq/g*/...): VALU / LDS busy share of the SIMD
cycles, wait shares per wave, LDS bank-conflict share.  SQ_ACTIVE_INST_* count quad-cycles per SIMD; durations come from a
<tag>_kernel_times.json (rocprofv3 stats of the same build); clock 2.4 GHz assumed (the shares scale with it).
usage: python tools/pmc_sq_table.py gpurun_out/pmcsq profiles/r03e_kernel_times.json > profiles/r03e_pmc_sq_summary.csv"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocprof_summary import stage
root, times = sys.argv[1], json.load(open(sys.argv[2]))["kernels"]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(root, "g*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        s = stage(r["Kernel_Name"])
        if s:
            acc[s][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[s][r["Counter_Name"]] += 1
g = lambda s, c: acc[s][c] / max(cnt[s][c], 1)
w = csv.writer(sys.stdout)
w.writerow(["stage", "avg_ms_rocprof", "valu_busy_share_of_simd_cycles", "lds_busy_share", "wave_cycles_waiting_any", "wave_cycles_waiting_issue",
            "lds_bank_conflict_share_of_lds_active", "valu_insts_per_wave", "lds_insts_per_wave", "vmem_rd_per_wave", "vmem_wr_per_wave", "waves"])
for s in sorted(acc, key=lambda s: -g(s, "SQ_BUSY_CYCLES")):
    if s not in times or not (s.startswith("kA") or s.startswith("kB")):
        continue
    ms = times[s]["avg_ms"]; simd_cycles = ms * 1e-3 * 2.4e9 * 1024; wc = max(g(s, "SQ_WAVE_CYCLES"), 1); wv = max(g(s, "SQ_WAVES"), 1)
    w.writerow([s, round(ms, 4), round(4 * g(s, "SQ_ACTIVE_INST_VALU") / simd_cycles, 3), round(4 * g(s, "SQ_ACTIVE_INST_LDS") / simd_cycles, 3),
                round(g(s, "SQ_WAIT_ANY") / wc, 3), round(g(s, "SQ_WAIT_INST_ANY") / wc, 3),
                round(g(s, "SQ_LDS_BANK_CONFLICT") / max(g(s, "SQ_LDS_IDX_ACTIVE"), 1), 3), round(g(s, "SQ_INSTS_VALU") / wv), round(g(s, "SQ_INSTS_LDS") / wv),
                round(g(s, "SQ_INSTS_VMEM_RD") / wv), round(g(s, "SQ_INSTS_VMEM_WR") / wv), round(wv)])
