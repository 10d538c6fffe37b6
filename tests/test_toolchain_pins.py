"""Two C++ standard-library semantics the oracle assumes for the reference's Eigen expressions (oracle/RECALLED.md rows 16, 18),
checked against the image's own g++ / libstdc++ / glibc -- the reference's toolchain family (CMakeLists.txt:27-33) -- on the CPU:
tests/cpp/toolchain_pins.cpp.  OpenCV's and Eigen's own code stay recalled (neither library is in the image)."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_libstdcxx_pow_and_complex_division(tmp_path):
    exe = str(tmp_path / "toolchain_pins")
    subprocess.run(["g++", "-O3", "-std=c++14", "-o", exe, os.path.join(ROOT, "tests", "cpp", "toolchain_pins.cpp")], check=True, timeout=300)
    d = json.loads(subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout)
    # Array::pow(int) (correlation_flow.cc:213): std::pow(float, int) returns double (a static_assert in the program) and, rounded to
    # float, equals BOTH the oracle's (float)pow((double)x, 3.0) and the HIP kernels' (float)((double)x*x*x) on every sample --
    # while a float-only cube does not (the promotion is real: it changes a quarter of the values)
    assert d["pow_samples"] >= 5000000 and d["libpow_ne_oracle"] == 0 and d["libpow_ne_double_cube"] == 0
    assert d["libpow_ne_float_cube"] > d["pow_samples"] // 10
    # complex<float> division (correlation_flow.cc:171): libstdc++'s operator/ and the textbook formula the oracle and the ridge
    # solve use are NOT bit-identical -- they differ in half the samples -- but never by more than 2 ulps: five orders of
    # magnitude below the measured float32 conditioning of T / (Kzz + lambda) (6e-4 of the peak), which the parity rule
    # already carries
    assert d["div_samples"] >= 5000000 and d["div_worst_ulps"] <= 2
    # fft_result.abs() (correlation_flow.cc:92): std::abs(complex<float>) (hypot) against sqrtf(re^2 + im^2): 15 % of the samples
    # differ, never by more than 1 ulp
    assert d["abs_samples"] >= 5000000 and d["abs_worst_ulps"] <= 1


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_pow_exponent_promotion_is_an_open_question_of_one_ulp(tmp_path):
    """RECALLED row 16, the OTHER reading: Eigen 3.3's member pow may promote an integer exponent to the array's
    scalar (promote_scalar_arg) and call powf(x, 3.0f).  With this image's glibc that differs from the double evaluation the
    oracle and the HIP kernels use on a fraction < 1e-3 of the samples, by one ulp -- of a plane that is normalised by its
    maximum and transformed next: no index moves.  Both the oracle (ora_set_pow_mode) and the kernels (-DKCC_POLY_POWF) carry
    the switch for whoever pins it with oracle/pin/."""
    src = tmp_path / "p.c"
    src.write_text(r'''
#include <math.h>
#include <stdio.h>
#include <stdint.h>
static uint64_t s = 88172645463325252ULL;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(void) {
    long diff = 0, n = 20000000L; int worst = 0;
    for (long i = 0; i < n; ++i) {
        const double u = (double)(rnd() >> 11) / 9007199254740992.0;
        const float mag = (float)exp(log(1e-3) + u * (log(2e5) - log(1e-3)));
        const float b = (rnd() & 1) ? mag : -mag;
        const float a = powf(b, 3.0f), c = (float)((double)b * (double)b * (double)b);
        if (a != c) { ++diff; const int ulps = (int)lrintf(fabsf(a - c) / (nextafterf(fabsf(c), INFINITY) - fabsf(c))); if (ulps > worst) worst = ulps; }
    }
    printf("{\"n\": %ld, \"diff\": %ld, \"worst_ulps\": %d}\n", n, diff, worst);
    return 0;
}''')
    exe = str(tmp_path / "p")
    subprocess.run(["gcc", "-O2", "-o", exe, str(src), "-lm"], check=True, timeout=300)
    d = json.loads(subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout)
    assert d["diff"] < d["n"] // 1000 and d["worst_ulps"] <= 1, d


def test_oracle_pow_switch_moves_no_index():
    """the oracle under both readings of Array::pow(int): the kernel planes differ in single bits, poses and arg-max indices of
    the golden-style pairs do not"""
    import numpy as np
    import sys
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import synth
    from oracle import kcc_oracle as ko
    cfg = ko.default_config(rotation_divisor=120, rotation_channel=80)
    keys, curs, _ = synth.make_batch(6, 60, 80, seed0=21, max_shift=8)
    L = ko.lib()
    try:
        L.ora_set_pow_mode(0)
        p0, i0, d0, _ = ko.track_pairs(cfg, keys, curs, True, nthreads=1)
        L.ora_set_pow_mode(1)
        assert L.ora_get_pow_mode() == 1
        p1, i1, d1, _ = ko.track_pairs(cfg, keys, curs, True, nthreads=1)
    finally:
        L.ora_set_pow_mode(0)
    assert np.array_equal(np.asarray(p0), np.asarray(p1))
    assert np.allclose(np.asarray(i0), np.asarray(i1), rtol=1e-4)
