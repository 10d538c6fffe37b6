"""Static instruction mix of the kernels in a gfx950 assembly file:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --offload-device-only kcc_kernels.hip -o kk.s
   python tools/isa_hist.py kk.s 'kA_inv<360, 1>' ...   (straight-line kernels: static counts = executed counts)"""
import collections, re, subprocess, sys
src = open(sys.argv[1]).read().splitlines()
want = sys.argv[2:]
funcs = {}; cur = None
for ln in src:
    m = re.match(r"^(_Z\w+):", ln)
    if m: cur = m.group(1); funcs[cur] = []; continue
    if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"): cur = None
    if cur and ln.startswith("\t") and not ln.startswith("\t."):
        op = ln.split()[0]
        if op.startswith(";"): continue
        funcs[cur].append(op)
names = subprocess.run(["c++filt"], input="\n".join(funcs), capture_output=True, text=True).stdout.splitlines()
def cls(op):
    if op.startswith(("v_fma", "v_fmac", "v_pk_fma")): return "fp fma"
    if op.startswith(("v_mul_f", "v_pk_mul")): return "fp mul"
    if op.startswith(("v_add_f", "v_sub_f", "v_subrev_f", "v_pk_add")): return "fp add"
    if op.startswith(("v_mov", "v_accvgpr")): return "mov"
    if op.startswith("v_cndmask") or op.startswith("v_cmp"): return "cmp/sel"
    if op.startswith(("v_exp", "v_rcp", "v_sqrt", "v_rsq", "v_log", "v_cvt", "v_max", "v_min", "v_rndne", "v_fract", "v_ldexp")): return "fp other"
    if op.startswith("v_"): return "int/addr"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"
for mangled, n in zip(funcs, names):
    short = n.replace("void kcc::", "").split("(")[0]
    if want and short not in want: continue
    if not want and not ("kA_" in short or "kB<" in short): continue
    h = collections.Counter(cls(o) for o in funcs[mangled])
    valu = sum(v for k, v in h.items() if k in ("fp fma", "fp mul", "fp add", "mov", "cmp/sel", "fp other", "int/addr"))
    print("%-24s VALU %5d: " % (short, valu) + "  ".join("%s %d" % (k, h[k]) for k in ("fp fma", "fp mul", "fp add", "fp other", "mov", "cmp/sel", "int/addr", "lds", "vmem", "waitcnt", "barrier", "salu")))
