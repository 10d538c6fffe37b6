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
