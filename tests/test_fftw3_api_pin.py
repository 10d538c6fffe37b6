"""Oracle FFT/IFFT pinned through the FFTW3 entry points the reference itself calls.

CorrelationFlow::FFT / IFFT (correlation_flow.cc:53-77) are `fftwf_plan_dft_r2c_2d(cols, rows, ...)`, `fftwf_plan_dft_c2r_2d(cols,
rows, ...)`, `fftwf_execute`, `fftwf_destroy_plan` and a division by x.size().  libfftw3f itself is not in this image, but Intel MKL
ships the FFTW3 *interface* (the same four symbols, same argument order, same r2c/c2r layout conventions) inside libmkl_rt, and an
Anaconda tree under /opt/conda carries it.  This test makes exactly the reference's calls on the reference's column-major buffers and
compares the oracle's ora_fft / ora_ifft with what comes back: a third independent float32 FFT (after hipFFT and numpy's float64
pocketfft), and the only one driven through the reference's own API -- so the (n0, n1) = (cols, rows) order, the halved axis and the
unnormalised forward / x.size() inverse are checked against a library's reading of those calls, not against our reading of them.

CPU-only; skipped when no libmkl_rt is found (the GPU box does not need it).  Runs in a child process so MKL's threading runtime
never shares a process with torch's.
"""
import ctypes as C
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

FFTW_ESTIMATE = 1 << 6
GEOMS = [(60, 80), (480, 640), (720, 480), (240, 360), (64, 720), (62, 94), (480, 752), (512, 512)]


def _find_mkl():
    for pat in ("/opt/conda/lib/libmkl_rt.so*", "/usr/lib/x86_64-linux-gnu/libmkl_rt.so*", "/opt/intel/oneapi/mkl/latest/lib/libmkl_rt.so*"):
        hits = sorted(glob.glob(pat))
        if hits:
            return hits[0]
    return None


def _worker(path):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import kcc_oracle as ko
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for f in (L.fftwf_plan_dft_r2c_2d, L.fftwf_plan_dft_c2r_2d):
        f.restype = C.c_void_p
        f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint]
    L.fftwf_execute.argtypes = [C.c_void_p]
    L.fftwf_destroy_plan.argtypes = [C.c_void_p]
    rng = np.random.default_rng(5)
    rows_out = []
    for rows, cols in GEOMS:
        orc = ko.Oracle(ko.default_config(), rows, cols)
        # ArrayXXf x(rows, cols), column-major: numpy (cols, rows) C-order holds the same bytes
        x = rng.random((cols, rows), dtype=np.float32)
        hr = rows // 2 + 1
        xf = np.zeros((cols, hr), np.complex64)                                   # ArrayXXcf xf(rows/2+1, cols)   :55
        xin = x.copy()
        p = L.fftwf_plan_dft_r2c_2d(cols, rows, xin.ctypes.data, xf.ctypes.data, FFTW_ESTIMATE)        # :56-57
        assert p, "MKL's FFTW3 interface refused the r2c plan"
        L.fftwf_execute(p); L.fftwf_destroy_plan(p)                               # :59-60
        ref64 = np.fft.rfft2(x.astype(np.float64))
        big = np.abs(ref64).max()
        e_fwd = float(np.abs(orc.fft(x) - xf).max() / big)
        e_lib = float(np.abs(xf - ref64).max() / big)
        # IFFT: private copy (c2r may destroy its input, :68), c2r, / x.size()   :70-76
        cxf = xf.copy()
        back = np.zeros((cols, rows), np.float32)
        p = L.fftwf_plan_dft_c2r_2d(cols, rows, cxf.ctypes.data, back.ctypes.data, FFTW_ESTIMATE)
        assert p, "MKL's FFTW3 interface refused the c2r plan"
        L.fftwf_execute(p); L.fftwf_destroy_plan(p)
        back = back / np.float32(rows * cols)
        e_inv = float(np.abs(orc.ifft(xf) - back).max())
        e_rt = float(np.abs(back - x).max())
        # the c2r rule on the self-conjugate bins: imaginary parts of DC / Nyquist rows are ignored, not an error
        bad = xf.copy()
        bad[0, 0] += 3j; bad[0, hr - 1] -= 2j; bad[cols // 2, 0] += 1j if cols % 2 == 0 else 0
        cb = bad.copy(); b2 = np.zeros((cols, rows), np.float32)
        p = L.fftwf_plan_dft_c2r_2d(cols, rows, cb.ctypes.data, b2.ctypes.data, FFTW_ESTIMATE)
        L.fftwf_execute(p); L.fftwf_destroy_plan(p)
        b2 = b2 / np.float32(rows * cols)
        e_c2r = float(np.abs(orc.ifft(bad) - b2).max())
        rows_out.append(dict(rows=rows, cols=cols, fwd=e_fwd, lib_vs_f64=e_lib, inv=e_inv, roundtrip=e_rt, c2r_rule=e_c2r))
    print("PIN " + json.dumps(rows_out))


def test_oracle_fft_against_fftw3_interface():
    path = _find_mkl()
    if path is None:
        pytest.skip("no libmkl_rt (FFTW3 interface) in this image")
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", MKL_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("PIN ")]
    assert line, r.stdout[-2000:]
    res = json.loads(line[0][4:])
    assert len(res) == len(GEOMS)
    for g in res:
        tag = "%dx%d: %s" % (g["rows"], g["cols"], g)
        # float32 FFTs of up to 3.6e5 points: a few ulps of the largest bin (same bound as the hipFFT pin)
        assert g["fwd"] < 2e-6, tag
        assert g["inv"] < 2e-6 and g["roundtrip"] < 2e-6, tag
        assert g["c2r_rule"] < 2e-6, tag


if __name__ == "__main__":
    _worker(sys.argv[1])
