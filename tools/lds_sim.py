"""Bank-conflict model of the FFT engine's LDS exchange (kcc_fft2.h fft_chain) for one plan / direction / workgroup shape.
ds_read_b64: 2 groups of 32 lanes, 64 dword banks; ds_write_b64: 4 groups of 16 lanes, 64 banks (MI355X_MICROARCH.md +
own measurement).  Cost of a group = max over banks of the number of DISTINCT dwords hitting that bank (>= 1 if any lane active).
usage: python tools/lds_sim.py N R1 R2 [R3] --lines L [--inv] [--nv 1]"""
import argparse, collections

def padc(rf): return 1 if rf % 2 == 0 else 2

def sim(N, radices, inv, lines, nv=1, rgroup=32, wgroup=16, verbose=True):
    R = list(radices)
    NP = len(R)
    T = max(N // r for r in R)
    RF = (R[-1] if inv else R[0]); RM = R[1] if NP == 3 else 1; RL = (R[0] if inv else R[-1])
    MF, ML = N // RF, N // RL; MM = N // RM if NP == 3 else 0
    PAD, PADC = RF, padc(RF)
    ext = max(N + padc(R[0]) * (N // R[0]), N + padc(R[-1]) * (N // R[-1])) + 2
    EP = ((ext + 31 - (T % 32)) // 32) * 32 + (T % 32)
    phys = lambda i: i + PADC * (i // PAD)
    SL = ML + PADC * (ML // PAD); SM = (MM + PADC * (MM // PAD)) if NP == 3 else 0
    NT = lines * T
    def cost(addr_of, active, group):
        tot = 0; ideal = 0
        for w0 in range(0, NT, 64):
            for g0 in range(w0, min(w0 + 64, NT), group):
                banks = collections.defaultdict(set); any_ = False
                for tid in range(g0, min(g0 + group, NT)):
                    lk, j = divmod(tid, T)
                    if not active(j): continue
                    any_ = True
                    a = addr_of(lk, j)          # float2 index -> dwords 2a, 2a+1
                    for d in (2 * a, 2 * a + 1): banks[d % 64].add(d)
                if any_:
                    tot += max(len(s) for s in banks.values()); ideal += 1
        return tot, ideal
    rep = []
    def add(name, tot_ideal, count):
        t, i = tot_ideal; rep.append((name, count, t / max(i, 1)))
    for v in range(nv):
        base = lambda lk, v=v: (nv * lk + v) * EP
        add("pass1 write", (lambda: (sum(cost(lambda lk, j, q=q: base(lk) + j * (RF + PADC) + q, lambda j: j < MF, wgroup)[0] for q in range(RF)),
                                      sum(cost(lambda lk, j, q=q: base(lk) + j * (RF + PADC) + q, lambda j: j < MF, wgroup)[1] for q in range(RF))))(), RF)
        if NP == 3:
            add("pass2 read", (sum(cost(lambda lk, j, q=q: base(lk) + phys(j) + q * SM, lambda j: j < MM, rgroup)[0] for q in range(RM)),
                               sum(cost(lambda lk, j, q=q: base(lk) + phys(j) + q * SM, lambda j: j < MM, rgroup)[1] for q in range(RM))), RM)
            add("pass2 write", (sum(cost(lambda lk, j, q=q: base(lk) + (j // RF) * (RF * RM + PADC * RM) + j % RF + q * (RF + PADC), lambda j: j < MM, wgroup)[0] for q in range(RM)),
                                sum(cost(lambda lk, j, q=q: base(lk) + (j // RF) * (RF * RM + PADC * RM) + j % RF + q * (RF + PADC), lambda j: j < MM, wgroup)[1] for q in range(RM))), RM)
        add("last read", (sum(cost(lambda lk, j, q=q: base(lk) + phys(j) + q * SL, lambda j: j < ML, rgroup)[0] for q in range(RL)),
                          sum(cost(lambda lk, j, q=q: base(lk) + phys(j) + q * SL, lambda j: j < ML, rgroup)[1] for q in range(RL))), RL)
        break
    if verbose:
        print("N=%d plan=%s %s T=%d lines=%d EPITCH=%d PADC=%d" % (N, R, "inv" if inv else "fwd", T, lines, EP, PADC))
        for name, cnt, mult in rep: print("   %-12s x%-3d conflict multiplier %.2f" % (name, cnt, mult))
    return rep

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("N", type=int); ap.add_argument("radices", type=int, nargs="+")
    ap.add_argument("--lines", type=int, default=4); ap.add_argument("--inv", action="store_true"); ap.add_argument("--nv", type=int, default=1)
    a = ap.parse_args()
    sim(a.N, a.radices, a.inv, a.lines, a.nv)
