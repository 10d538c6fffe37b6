"""A THIRD-PARTY FFT under the oracle (allowed in tests by SURVEY 7): hipFFT -- AMD's FFT library, no
code shared with this repository -- runs the reference's own transform shape on the GPU box,

    fftwf_plan_dft_r2c_2d(n0 = cols, n1 = rows)   /   fftwf_plan_dft_c2r_2d   (src/correlation_flow.cc:56-61, 70-75),

through its C API (hipfftPlan2d / hipfftExecR2C / hipfftExecC2R), and BOTH FFTs of this repository are compared with it on
the same inputs: the oracle's (oracle/kcc_oracle.c, the checker of every parity test) and the product's HIP engine
(nik_dbg_fft / nik_dbg_ifft).  It pins what "FFT" and "IFFT" mean -- layout (the halved axis is the row axis, output
(rows/2+1) x cols column-major), sign, normalisation (forward unnormalised, inverse / size) and float32 accuracy -- to something
neither side wrote.  (FFTW itself is not in the image; hipFFT implements the same FFTW r2c / c2r conventions.)

GPU only: hipFFT needs a device.  The product path never touches hipFFT (tests/test_abi.py::test_no_oracle_on_product_path
and `readelf -d` show no such dependency)."""
import ctypes as C

import numpy as np
import pytest

from kcc_helpers import FULL, SMALL, nik
from oracle import kcc_oracle as ko

pytestmark = pytest.mark.gpu

HIPFFT_R2C, HIPFFT_C2R = 0x2A, 0x2C


def _hipfft():
    L = C.CDLL("libhipfft.so")
    L.hipfftPlan2d.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
    L.hipfftExecR2C.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hipfftExecC2R.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hipfftDestroy.argtypes = [C.c_void_p]
    # device buffers through the HIP runtime hipFFT itself links (no torch in this test: two HIP runtimes in one process --
    # torch bundles its own -- did not both find the GPU once hipFFT was loaded)
    R = C.CDLL("libamdhip64.so")
    R.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    R.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    R.hipFree.argtypes = [C.c_void_p]
    L.rt = R
    return L


def _run(L, kind, src, out_shape, out_dtype, cols, rows):
    R = L.rt
    src = np.ascontiguousarray(src)
    out = np.empty(out_shape, out_dtype)
    d_in, d_out, plan = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert R.hipMalloc(C.byref(d_in), src.nbytes) == 0 and R.hipMalloc(C.byref(d_out), out.nbytes) == 0
    assert R.hipMemcpy(d_in, src.ctypes.data_as(C.c_void_p), src.nbytes, 1) == 0
    assert L.hipfftPlan2d(C.byref(plan), cols, rows, kind) == 0
    assert (L.hipfftExecR2C if kind == HIPFFT_R2C else L.hipfftExecC2R)(plan, d_in, d_out) == 0
    assert R.hipDeviceSynchronize() == 0
    assert R.hipMemcpy(out.ctypes.data_as(C.c_void_p), d_out, out.nbytes, 2) == 0
    L.hipfftDestroy(plan); R.hipFree(d_in); R.hipFree(d_out)
    return out


def _hipfft_r2c(L, x):
    """x: numpy (cols, rows) float32 = the column-major rows x cols array handed to FFTW as row-major cols x rows"""
    cols, rows = x.shape
    return _run(L, HIPFFT_R2C, x.astype(np.float32), (cols, rows // 2 + 1), np.complex64, cols, rows)


def _hipfft_c2r(L, xf):
    cols, hr = xf.shape
    rows = (hr - 1) * 2                                                      # (c2r may destroy its input: _run uploads a private copy)
    return _run(L, HIPFFT_C2R, xf.astype(np.complex64), (cols, rows), np.float32, cols, rows) / np.float32(rows * cols)   # IFFT: x / x.size() (correlation_flow.cc:76)


def _relmax(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("geom", [pytest.param(SMALL, id="60x80"), pytest.param(FULL, id="480x640")])
def test_oracle_and_hip_fft_against_hipfft(geom):
    N = nik()
    L = _hipfft()
    cfg = N.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"])
    cf = N.CorrelationFlow(cfg, geom["H"], geom["W"], max_batch=2, max_frames=4)
    orc = ko.Oracle(ko.default_config(rotation_divisor=geom["PD"], rotation_channel=geom["PC"]), geom["H"], geom["W"])
    rng = np.random.default_rng(11)
    for which, (rows, cols) in enumerate([(geom["H"], geom["W"]), (geom["PD"], geom["PC"])]):
        x = rng.random((cols, rows), dtype=np.float32)
        third = _hipfft_r2c(L, x)
        ref64 = np.fft.rfft2(x.astype(np.float64))                           # (context: how far float32 FFTs sit from the exact one)
        e_third = _relmax(third, ref64)
        e_orc, e_hip = _relmax(orc.fft(x), third), _relmax(cf.dbg_fft(x, which), third)
        # three independent float32 FFTs of 3e5 points agree to a few ulps of the largest bin
        assert e_orc < 2e-6 and e_hip < 2e-6, "forward (which=%d): oracle %.2e, HIP %.2e from hipFFT (hipFFT %.2e from float64)" % (which, e_orc, e_hip, e_third)
        # inverse, from a Hermitian-consistent spectrum (the only kind the path produces)
        back3 = _hipfft_c2r(L, third)
        b_orc, b_hip = np.abs(orc.ifft(third) - back3).max(), np.abs(cf.dbg_ifft(third, which) - back3).max()
        assert b_orc < 2e-6 and b_hip < 2e-6 and np.abs(back3 - x).max() < 2e-6, "inverse (which=%d): oracle %.2e, HIP %.2e from hipFFT" % (which, b_orc, b_hip)
    # known answers through the third party as well: the target spectrum of the unit impulse at the centre is (-1)^(k+l)
    H, W = geom["H"], geom["W"]
    d = np.zeros((W, H), np.float32); d[W // 2, H // 2] = 1
    l, k = np.meshgrid(np.arange(W), np.arange(H // 2 + 1), indexing="ij")
    assert np.abs(_hipfft_r2c(L, d) - ((-1.0) ** (k + l))).max() < 1e-5
    cf.close()


def test_zero_phase_image_against_hipfft():
    """IFFT(|FFT(img)|) -- the first half of ComputeIntermedium (correlation_flow.cc:91-92) -- evaluated with hipFFT only and
    compared with the oracle's: the even, translation-invariant plane the polar transform samples"""
    L = _hipfft()
    g = SMALL
    orc = ko.Oracle(ko.default_config(rotation_divisor=g["PD"], rotation_channel=g["PC"]), g["H"], g["W"])
    x = np.random.default_rng(5).random((g["W"], g["H"]), dtype=np.float32)
    F = _hipfft_r2c(L, x)
    p3 = _hipfft_c2r(L, np.abs(F).astype(np.complex64))
    po = orc.ifft(np.abs(orc.fft(x)).astype(np.complex64))
    assert np.abs(po - p3).max() < 2e-6 * np.abs(p3).max()
    # even symmetry p(r, c) = p(-r, -c): the property the mirrored half-plane kernels rely on
    assert np.abs(p3 - np.roll(p3[::-1, ::-1], (1, 1), axis=(0, 1))).max() < 1e-5 * np.abs(p3).max()
