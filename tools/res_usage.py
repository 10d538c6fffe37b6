"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output for the FFT kernels:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage -c kcc_kernels.hip -o /tmp/kk.o 2> log
   python tools/res_usage.py log [regex]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"<(240|360|480|640)\b")
cur = None; recs = {}
for line in txt.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m: cur = m.group(1); recs[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
    if m and cur: recs[cur][m.group(1).strip()] = m.group(2)
names = subprocess.run(["c++filt"], input="\n".join(recs), capture_output=True, text=True).stdout.splitlines()
for mangled, n in zip(recs, names):
    r = recs[mangled]
    if ("kA_" in n or "kB<" in n) and pat.search(n):
        lds = int(r.get("LDS Size [bytes/block]", 0))
        print("%-50s vgpr %3s agpr %3s scratch %3s occ(waves/SIMD by regs) %s lds %6d -> %d WG/CU by LDS" % (
            n.replace("void kcc::", "")[:50], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), lds, 163840 // max(lds, 1)))
