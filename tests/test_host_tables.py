"""CPU replay of the two LDS-staged gather kernels' table logic against the oracle (no GPU).

The HIP kernels kA_fwd<., SRC_POLAR_*> and kA_fwd<., SRC_ROT8> consume tables built on the host
(ni-slam_amd/csrc/kcc_tables.cpp) and do only staging + bilinear sampling.  These tests rebuild the same tables
through the C ABI's host hooks and replay the kernels' arithmetic in numpy (same float32 operation order), so a
table bug shows up here, bit for bit, before any GPU run:
  * polar: staging chunks -> LDS image -> per-thread samples  == oracle polar(fftshift(remove_zero(x)))
  * de-rotation: band boxes (wrap applied while staging) -> byte taps -> /255 -> bilerp == oracle rotate(x, deg)
  * unit_u8(v) == float32(v) / float32(255) for all 256 values
  * the compile-time box bounds hold for every 0.5-degree angle
"""
from fractions import Fraction as F

import os

import numpy as np
import pytest

import synth
from kcc_helpers import FULL, ROOT, SMALL, nik
from oracle import kcc_oracle as ko

GEOMS = [pytest.param(SMALL, id="60x80"), pytest.param(FULL, id="480x640")]
f32 = np.float32


def bilerp(v0, v1, v2, v3, fx, fy):
    """kcc_kernels.hip bilerp(): cv::remap's float weights, summed in OpenCV's order, float32 throughout"""
    s = f32(1.0 / 32.0)
    tx1 = fx.astype(f32) * s
    tx0 = f32(1) - tx1
    ty1 = fy.astype(f32) * s
    ty0 = f32(1) - ty1
    w0, w1, w2, w3 = ty0 * tx0, ty0 * tx1, ty1 * tx0, ty1 * tx1
    acc = v0 * w0
    acc = acc + v1 * w1
    acc = acc + v2 * w2
    acc = acc + v3 * w3
    return acc


def _rn32(fr):
    if fr == 0:
        return f32(0)
    x = f32(float(fr))
    cands = [x, np.nextafter(x, f32(np.inf)), np.nextafter(x, f32(-np.inf))]
    return f32(min(cands, key=lambda c: (abs(F(float(c)) - fr), int(f32(c).view(np.uint32)) & 1)))


def test_unit_u8_is_exact_division():
    """unit_u8(v) = fma(v, chi, RN(v*clo)) equals the IEEE float32 division v/255 for every byte"""
    chi, clo = F(float.fromhex("0x1.010102p-8")), F(float.fromhex("-0x1.fdfdfep-33"))
    for v in range(256):
        t = _rn32(F(v) * clo)
        q = _rn32(F(v) * chi + F(float(t)))
        assert q == f32(v) / f32(255.0), v


def _shifted_plane(x):
    """S[W+1][H+2]: fftshift(remove_zero(x)) with the zero border the gather relies on"""
    W, H = x.shape
    hp = ko.Oracle.fftshift(ko.Oracle.remove_zero(x))
    S = np.zeros((W + 1, H + 2), f32)
    S[:W, :H] = hp
    return S


@pytest.mark.parametrize("geom", GEOMS)
def test_polar_plan_replay_matches_oracle(geom):
    H, W, PD, PC = geom["H"], geom["W"], geom["PD"], geom["PC"]
    N = nik()
    plan = N.host_polar_plan(H, W, PD, PC)
    qs, nseg, tiles, lines, threads, rf, mf = (plan[k] for k in ("qs", "nseg", "tiles", "lines", "threads", "rf", "mf"))
    assert rf * mf == PD // 2 and tiles * lines == PC and qs * nseg == rf
    x = np.random.default_rng(2).random((W, H), dtype=f32)
    S = np.concatenate([_shifted_plane(x).ravel(), np.zeros(16, f32)])      # (+16: the last chunk may run past the plane)
    orc = ko.Oracle(ko.default_config(rotation_divisor=PD, rotation_channel=PC), H, W)
    ref = orc.polar(ko.Oracle.fftshift(ko.Oracle.remove_zero(x)))          # (PC, PD): ref[rho, phi]
    got = np.full((PC, PD), np.nan, f32)
    worst = 0
    for t in range(tiles):
        for s in range(nseg):
            c0, c1 = plan["seg_first"][t * nseg + s], plan["seg_first"][t * nseg + s + 1]
            d = plan["chunks"][c0:c1]
            worst = max(worst, len(d))
            # LDS image: chunk c -> floats [16c, 16c+16) = 16 consecutive source floats from its offset
            off = d.astype(np.int64)[:, None] + np.arange(16)[None, :]
            assert off.max() < S.size
            lds = S[off].ravel()
            for qq in range(qs):
                q = s * qs + qq
                e = plan["pts"][t, q].reshape(lines, threads, 4)[:, :mf, :]
                for h in range(2):
                    lo, hi = e[..., 2 * h].astype(np.int64), e[..., 2 * h + 1].astype(np.int64)
                    oa, ob = lo & 0xFFFF, hi
                    assert max(oa.max(), ob.max()) + 1 < lds.size
                    val = bilerp(lds[oa], lds[ob], lds[oa + 1], lds[ob + 1], (lo >> 16) & 31, (lo >> 21) & 31)
                    phi = 2 * (np.arange(mf)[None, :] + q * mf) + h
                    got[t * lines + np.arange(lines)[:, None], phi] = val
    assert worst * 64 == plan["lds_bytes"] and plan["lds_bytes"] <= 150 * 1024
    assert np.array_equal(got, ref)


def _replay_rot8(img, deg, geom8, terms):
    """kA_fwd<., SRC_ROT8>: every band's bounding box staged with BORDER_WRAP, byte taps at un-wrapped coordinates"""
    H, W = img.shape
    ad, bd, X0, Y0 = (t.astype(np.int64) for t in terms)
    BR, NB, BH, PITCH = geom8["band_rows"], geom8["bands"], geom8["box_rows"], geom8["pitch"]
    assert BR * NB == H
    padded = np.concatenate([img, img[:, :16]], axis=1)                    # frame store row: columns 0..15 repeated
    out = np.zeros((W, H), f32)                                             # column-major like the oracle
    unit = np.arange(256, dtype=f32) / f32(255.0)
    for x0 in range(0, W, 16):
        for b in range(NB):
            r0, r1 = b * BR, b * BR + BR - 1
            xmin = (min(X0[r0], X0[r1]) + min(ad[x0], ad[x0 + 15])) >> 10
            ymin = (min(Y0[r0], Y0[r1]) + min(bd[x0], bd[x0 + 15])) >> 10
            ox, oy = xmin & ~3, ymin
            ys = oy + np.arange(BH)
            ys = np.where(ys < 0, ys + H, ys); ys = np.where(ys >= H, ys - H, ys)
            xs = ox + 16 * np.arange(PITCH // 16)
            xs = np.where(xs < 0, xs + W, xs); xs = np.where(xs >= W, xs - W, xs)
            assert ((ys >= 0) & (ys < H)).all() and ((xs >= 0) & (xs < W)).all()
            box = np.stack([np.concatenate([padded[y, x:x + 16] for x in xs]) for y in ys])     # [BH][PITCH]
            r = np.arange(r0, r1 + 1)[:, None]; c = np.arange(x0, x0 + 16)[None, :]
            X = (X0[r] + ad[c]) >> 5; Y = (Y0[r] + bd[c]) >> 5
            bx, by = (X >> 5) - ox, (Y >> 5) - oy
            assert bx.min() >= 0 and bx.max() + 1 < PITCH and by.min() >= 0 and by.max() + 1 < BH, (deg, x0, b)
            v = bilerp(unit[box[by, bx]], unit[box[by, bx + 1]], unit[box[by + 1, bx]], unit[box[by + 1, bx + 1]], X & 31, Y & 31)
            out[x0:x0 + 16, r0:r1 + 1] = v.T
    return out


@pytest.mark.parametrize("geom", GEOMS)
def test_rot8_replay_matches_oracle(geom):
    H, W = geom["H"], geom["W"]
    N = nik()
    g8 = N.host_rot8_geom(H)
    img = synth.canvas(3, H, W)[:H, :W]
    x = ko.Oracle.normalize_u8(img)
    degs = (0.0, 0.5, -18.5, 44.5, 90.0, 179.5, -301.0) if geom is SMALL else (7.5, -58.5, 180.0)
    for deg in degs:
        got = _replay_rot8(img, deg, g8, N.host_rot_terms(H, W, deg))
        assert np.array_equal(got, ko.Oracle.rotate(x, deg)), deg


@pytest.mark.parametrize("H,W", [(60, 80), (120, 160), (240, 320), (480, 640), (448, 448), (720, 1280), (1200, 1600)])
def test_rot8_boxes_hold_for_every_angle(H, W):
    """the compile-time box (box_rows x pitch) contains every band's source rectangle for all 1440 half-degree angles"""
    N = nik()
    g8 = N.host_rot8_geom(H)
    BR, NB, BH, PITCH = g8["band_rows"], g8["bands"], g8["box_rows"], g8["pitch"]
    assert BR * NB == H and g8["lds_bytes"] <= 160 * 1024
    r0 = np.arange(NB) * BR; r1 = r0 + BR - 1
    c0 = np.arange(0, W, 16); c1 = c0 + 15
    for deg2 in range(-720, 721):
        ad, bd, X0, Y0 = (t.astype(np.int64) for t in N.host_rot_terms(H, W, deg2 * 0.5))
        # monotone terms: the extremes of X0[r] + ad[c] over a block sit at its corners
        assert (np.diff(X0) >= 0).all() or (np.diff(X0) <= 0).all()
        assert (np.diff(ad) >= 0).all() or (np.diff(ad) <= 0).all()
        assert (np.diff(Y0) >= 0).all() or (np.diff(Y0) <= 0).all()
        assert (np.diff(bd) >= 0).all() or (np.diff(bd) <= 0).all()
        xlo = (np.minimum(X0[r0], X0[r1])[:, None] + np.minimum(ad[c0], ad[c1])[None, :]) >> 10
        xhi = ((np.maximum(X0[r0], X0[r1])[:, None] + np.maximum(ad[c0], ad[c1])[None, :]) >> 10) + 1
        ylo = (np.minimum(Y0[r0], Y0[r1])[:, None] + np.minimum(bd[c0], bd[c1])[None, :]) >> 10
        yhi = ((np.maximum(Y0[r0], Y0[r1])[:, None] + np.maximum(bd[c0], bd[c1])[None, :]) >> 10) + 1
        assert (xhi - (xlo & ~3) < PITCH).all() and (yhi - ylo < BH).all(), deg2
        # single-step BORDER_WRAP while staging: every staged coordinate within one period of the image
        assert ((xlo & ~3) >= -W).all() and ((xlo & ~3) + PITCH <= 2 * W).all() and (ylo >= -H).all() and (ylo + BH <= 2 * H).all(), deg2


def test_exchange_swizzles_are_conflict_free_permutations():
    """kcc_fft2.h stores the unpadded exchange buffers of the 240- and 640-point plans at XOR-swizzled positions.  Each
    swizzle must be a permutation of [0, N) (nothing is overwritten) and, per tools/lds_sim3.py's bank model (64 dword banks,
    16-lane b64 writes, 32-lane b64 reads), every access of the plan must be conflict-free at the shipped workgroup shapes."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    from lds_sim3 import worst, epitch
    assert sorted(i ^ ((i >> 4) & 15) for i in range(240)) == list(range(240))
    assert sorted(i ^ ((i >> 5) & 3) for i in range(640)) == list(range(640))
    assert sorted(i ^ (((i >> 6) & 1) << 3) for i in range(640)) == list(range(640))
    # 640 = 8 x 8 x 10, T = 80 threads per line, 3 (two-plane kernels) or 4 lines per workgroup
    T, EP = 80, epitch(640, 80)
    assert EP == 656
    for lines in (3, 4):
        NT = lines * T
        acc = {
            "f w1": [(lambda lk, j, q=q: lk * EP + 8 * j + (q ^ ((j >> 2) & 3)), 80, 16) for q in range(8)],
            "f r2": [(lambda lk, j, q=q: lk * EP + ((j + 80 * q) ^ (((j + 80 * q) >> 5) & 3)), 80, 32) for q in range(8)],
            "f w2": [(lambda lk, j, q=q: lk * EP + ((64 * (j // 8) + j % 8 + 8 * q) ^ (((j // 8) & 1) << 3)), 80, 16) for q in range(8)],
            "f r3": [(lambda lk, j, q=q: lk * EP + ((j + 64 * q) ^ (8 * (q & 1))), 64, 32) for q in range(10)],
            "i w1": [(lambda lk, j, q=q: lk * EP + 10 * j + q, 64, 16) for q in range(10)],
            "i r2": [(lambda lk, j, q=q: lk * EP + j + 80 * q, 80, 32) for q in range(8)],
            "i w2": [(lambda lk, j, q=q: lk * EP + 80 * (j // 10) + j % 10 + 10 * q, 80, 16) for q in range(8)],
            "i r3": [(lambda lk, j, q=q: lk * EP + j + 80 * q, 80, 32) for q in range(8)],
        }
        for name, lst in acc.items():
            for f, active, group in lst:
                assert worst(f, lambda j, a=active: j < a, NT, T, group) == 1, (lines, name)
    # 240 = 16 x 15, T = 16, 16 lines per workgroup, line pitch 240
    T, EP, NT = 16, epitch(240, 16), 256
    assert EP == 240
    for q in range(16):
        assert worst(lambda lk, j, q=q: lk * EP + 16 * j + (q ^ (j & 15)), lambda j: j < 15, NT, T, 16) == 1
    for q in range(15):
        assert worst(lambda lk, j, q=q: lk * EP + 16 * q + (j ^ (q & 15)), lambda j: j < 16, NT, T, 32) == 1
    for q in range(15):                                       # inverse: first radix 15, plain
        assert worst(lambda lk, j, q=q: lk * EP + 15 * j + q, lambda j: j < 16, NT, T, 16) == 1
    for q in range(16):
        assert worst(lambda lk, j, q=q: lk * EP + j + 15 * q, lambda j: j < 15, NT, T, 32) <= 2


def test_fft_butterflies_on_the_host(tmp_path):
    """The in-register butterflies of the FFT engine (kcc_fft2.h dft_run: Good-Thomas prime-factor and Cooley-Tukey splits over
    the base radices of kcc_fft.h) compiled for the HOST (g++, tests/cpp/hipstub) and compared with a direct float64 DFT, both
    directions, every radix any plan uses; dft_pos must be a permutation."""
    import subprocess
    exe = os.path.join(str(tmp_path), "dft_host_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "tests", "cpp", "hipstub"),
                           "-I" + os.path.join(ROOT, "ni-slam_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "dft_host_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "pfa=1" in out.stdout and "pfa=0" in out.stdout


def test_run_time_fft_plans_cover_every_even_length():
    """the any-size kernels' plan of a line (nik_host_fft_plan, no GPU): for every even length the reference could hand over up to
    8192 the radices multiply back to the length, the in-register butterflies (16, 15, 12, 10, 9, 8 ... 2) come first and whatever
    is left is prime (it runs as a direct DFT pass)"""
    import ctypes as C
    N = nik()
    L = N.load()
    L.nik_host_fft_plan.argtypes = [C.c_int, C.POINTER(C.c_int)]
    r = (C.c_int * 16)()
    def is_prime(v):
        return v > 1 and all(v % d for d in range(2, int(v ** 0.5) + 1))
    for n in list(range(4, 8193, 2)) + [47, 61, 79, 6561, 8191]:
        k = L.nik_host_fft_plan(n, r)
        rad = list(r[:k])
        assert k >= 1 and int(np.prod(rad, dtype=np.int64)) == n, (n, rad)
        assert all(x in (16, 15, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2) or is_prime(x) for x in rad), (n, rad)
    assert list(r[:L.nik_host_fft_plan(752, r)]) == [16, 47] and list(r[:L.nik_host_fft_plan(480, r)]) == [16, 15, 2]
    assert list(r[:L.nik_host_fft_plan(720, r)]) == [16, 15, 3] and list(r[:L.nik_host_fft_plan(640, r)]) == [16, 10, 4]
