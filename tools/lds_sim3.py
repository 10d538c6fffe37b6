"""Search the row pitches of the 3-pass exchange layout (kcc_fft2.h, NP == 3):
   exchange 1: output q of first-pass butterfly j        at q*X1P + j            (written contiguously over j)
               read by second-pass butterfly j=(jb,k)    at k*X1P + jb + q*(MM/RF)
   exchange 2: output q of second-pass butterfly (jb,k)  at jb*X2P + k + q*RF
               read by last-pass butterfly j             at q*X2P + j            (contiguous over j)
so that every ds_write_b64 (16-lane groups) and ds_read_b64 (32-lane groups) of a workgroup is conflict-free.
usage: python tools/lds_sim3.py N R1 R2 R3 --lines L [--nv 1|2]"""
import argparse, collections

def epitch(ext, T): return ((ext + 31 - (T % 32)) // 32) * 32 + (T % 32)

def worst(addr_of, active, NT, T, group):
    """max conflict multiplier over the lane groups of the workgroup"""
    w = 1
    for g0 in range(0, NT, group):
        if g0 // 64 != (min(g0 + group, NT) - 1) // 64: pass
        banks = collections.defaultdict(set)
        for tid in range(g0, min(g0 + group, NT)):
            lk, j = divmod(tid, T)
            if not active(j): continue
            a = addr_of(lk, j)
            for d in (2 * a, 2 * a + 1): banks[d % 64].add(d)
        if banks: w = max(w, max(len(s) for s in banks.values()))
    return w

def check(N, R, inv, lines, nv, X1P, X2P, EP):
    T = max(N // r for r in R)
    RF, RM, RL = (R[2], R[1], R[0]) if inv else (R[0], R[1], R[2])
    MF, MM, ML = N // RF, N // RM, N // RL
    NT = lines * T
    for v in range(nv):
        base = lambda lk: (nv * lk + v) * EP
        for q in range(RF):
            if worst(lambda lk, j: base(lk) + q * X1P + j, lambda j: j < MF, NT, T, 16) > 1: return "w1"
        for q in range(RM):
            if worst(lambda lk, j: base(lk) + (j % RF) * X1P + j // RF + q * (MM // RF), lambda j: j < MM, NT, T, 32) > 1: return "r2"
        for q in range(RM):
            if worst(lambda lk, j: base(lk) + (j // RF) * X2P + j % RF + q * RF, lambda j: j < MM, NT, T, 16) > 1: return "w2"
        for q in range(RL):
            if worst(lambda lk, j: base(lk) + q * X2P + j, lambda j: j < ML, NT, T, 32) > 1: return "r3"
    return None

def search(N, R, lines, nv, verbose=True):
    T = max(N // r for r in R)
    best = None
    out = {}
    for inv in (False, True):
        RF, RM, RL = (R[2], R[1], R[0]) if inv else (R[0], R[1], R[2])
        MF = N // RF; RR = RF * RM
        found = None
        for X1P in range(MF, MF + 40):
            for X2P in range(RR, RR + 40):
                ext = max(RF * X1P, RL * X2P)
                # the buffer pitch is shared by both directions: accept any EP candidate here, fix up below
                EP = epitch(ext, T)
                if check(N, R, inv, lines, nv, X1P, X2P, EP) is None:
                    if found is None or ext < found[2]: found = (X1P, X2P, ext)
        out[inv] = found
        if verbose: print("N=%d %s %s lines=%d nv=%d: X1P=%s X2P=%s ext=%s" % (N, R, "inv" if inv else "fwd", lines, nv, *(found or (None, None, None))))
    return out

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("N", type=int); ap.add_argument("radices", type=int, nargs=3)
    ap.add_argument("--lines", type=int, default=4); ap.add_argument("--nv", type=int, default=1)
    a = ap.parse_args()
    search(a.N, a.radices, a.lines, a.nv)

def check2(N, R, inv, lines, nv, mode1, X1P, X2P, EP):
    """mode1 = 'plain' (exchange 1 at o = j*RF + q, unpadded) or 'transposed' (q*X1P + j)"""
    T = max(N // r for r in R)
    RF, RM, RL = (R[2], R[1], R[0]) if inv else (R[0], R[1], R[2])
    MF, MM, ML = N // RF, N // RM, N // RL
    NT = lines * T
    for v in range(nv):
        base = lambda lk: (nv * lk + v) * EP
        for q in range(RF):
            f = (lambda lk, j: base(lk) + j * RF + q) if mode1 == "plain" else (lambda lk, j: base(lk) + q * X1P + j)
            if worst(f, lambda j: j < MF, NT, T, 16) > 1: return "w1"
        for q in range(RM):
            f = (lambda lk, j: base(lk) + j + q * MM) if mode1 == "plain" else (lambda lk, j: base(lk) + (j % RF) * X1P + j // RF + q * (MM // RF))
            if worst(f, lambda j: j < MM, NT, T, 32) > 1: return "r2"
        for q in range(RM):
            if worst(lambda lk, j: base(lk) + (j // RF) * X2P + j % RF + q * RF, lambda j: j < MM, NT, T, 16) > 1: return "w2"
        for q in range(RL):
            if worst(lambda lk, j: base(lk) + q * X2P + j, lambda j: j < ML, NT, T, 32) > 1: return "r3"
    return None

def joint(N, R, configs):
    """configs: list of (lines, nv).  Prints, per direction, the (mode1, X1P, X2P) choices and the line-pitch residues mod 32
    that are conflict-free for every config, then the residues common to both directions."""
    T = max(N // r for r in R)
    per_dir = {}
    for inv in (False, True):
        RF, RM, RL = (R[2], R[1], R[0]) if inv else (R[0], R[1], R[2])
        MF = N // RF; RR = RF * RM
        cands = []
        for mode1 in ("plain", "transposed"):
            for X1P in ([0] if mode1 == "plain" else range(MF, MF + 33)):
                for X2P in range(RR, RR + 33):
                    ext = max(N if mode1 == "plain" else RF * X1P, RL * X2P)
                    ok = [e for e in range(32) if all(check2(N, R, inv, l, nv, mode1, X1P, X2P, ((ext + 31) // 32) * 32 + 32 + e) is None for (l, nv) in configs)]
                    if ok: cands.append((ext, mode1, X1P, X2P, ok))
        cands.sort(key=lambda c: c[0])
        per_dir[inv] = cands
        print("N=%d %s %s: %d layouts; smallest:" % (N, R, "inv" if inv else "fwd", len(cands)))
        for c in cands[:5]: print("    ext=%d mode1=%s X1P=%d X2P=%d  line pitch mod 32 in %s" % c)
    for cf in per_dir[False][:12]:
        for ci in per_dir[True][:12]:
            common = sorted(set(cf[4]) & set(ci[4]))
            if common: print("  joint: fwd(%s,%d,%d) inv(%s,%d,%d) ext=%d pitch mod 32 in %s" % (cf[1], cf[2], cf[3], ci[1], ci[2], ci[3], max(cf[0], ci[0]), common))
    return per_dir
