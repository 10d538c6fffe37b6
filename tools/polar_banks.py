"""Bank-conflict model of the polar gather's LDS taps (host only): replays the per-thread sample entries of the plan nik_create
builds (nik_host_polar_plan) wave by wave.  A tap is an 8-byte read at a 4-byte-aligned float index; per half-wave (32 lanes) the
cost is the largest number of distinct dwords any of the 32 banks has to deliver (the conflict-free count is 2).
usage: [NIK_POLAR_SKEW_K=k NIK_POLAR_SKEW_M=m] python tools/polar_banks.py [H W PD PC]"""
import os, sys, importlib.util
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("nk", os.path.join(R, "ni-slam_amd", "nislam_kcc.py")); N = importlib.util.module_from_spec(spec); spec.loader.exec_module(N)
H, W, PD, PC = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (480, 640, 720, 480)
p = N.host_polar_plan(H, W, PD, PC)
pts = p["pts"]                                  # [tiles][rf][NT][4]
tiles, rf, NT, _ = pts.shape
mf, threads, lines = p["mf"], p["threads"], p["lines"]
tid = np.arange(NT); active = (tid % threads) < mf
cost = ideal = 0
for t in range(0, tiles, max(1, tiles // 12)):                      # a sample of the tiles
    for q in range(rf):
        e = pts[t, q]
        for a in (e[:, 0] & 0xFFFF, e[:, 1], e[:, 2] & 0xFFFF, e[:, 3]):
            a = a.astype(np.int64)
            for w0 in range(0, NT, 32):
                sel = active[w0:w0 + 32]
                if not sel.any(): continue
                d = np.unique(np.concatenate([a[w0:w0 + 32][sel], a[w0:w0 + 32][sel] + 1]))
                cost += np.bincount(d % 32, minlength=32).max(); ideal += 2
nch = len(p["chunks"])
print("geometry %dx%d polar %dx%d: qs %d nseg %d lds %d B chunks %d  bank-cycle factor %.2f" % (H, W, PD, PC, p["qs"], p["nseg"], p["lds_bytes"], nch, cost / ideal))
