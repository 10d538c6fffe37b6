"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/nislam_kcc.h declares; without a GPU every entry point fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from kcc_helpers import PKG, ROOT, SMALL, load_module, nik


@pytest.fixture(scope="module")
def lib_path():
    build = load_module("nislam_build", os.path.join(PKG, "build.py"))
    return build.build()


def _declared():
    hdr = open(os.path.join(ROOT, "include", "nislam_kcc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(nik_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported(lib_path):
    L = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(L, n), "missing export %s" % n
    assert sorted(nik().EXPORTS) == names, "python binding and header disagree"


def test_config_struct_mirrors_cfconfig():
    N = nik()
    fields = [f[0] for f in N.NikConfig._fields_]
    assert fields == ["width", "height", "lambda_", "kernel", "sigma", "offset", "power", "rotation_divisor",
                      "rotation_channel"]                      # reference include/read_configs.h:15-25 order
    assert ctypes.sizeof(N.NikConfig) == 36
    assert ctypes.sizeof(N.NikPoseResult) == 3 * 8 + 3 * 8 + 4 * 2 + 4 * 4 + 4 * 3 + 4 + 4 + 4


def test_no_oracle_on_product_path():
    """the product path must not reach into oracle/ (a CPU fallback would void every parity claim)"""
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(root, f), errors="ignore").read()
                assert "kcc_oracle" not in src and "np_restatement" not in src, f


def test_fails_loudly_without_gpu(lib_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    N = nik()
    with pytest.raises(N.NikError) as e:
        N.CorrelationFlow(N.default_config(rotation_divisor=SMALL["PD"], rotation_channel=SMALL["PC"]), SMALL["H"], SMALL["W"])
    assert e.value.code == N.NIK_ERR_HIP and "no CPU fallback" in str(e.value)


def test_argument_validation_precedes_device_use(lib_path):
    N = nik()
    L = N.load()
    ctx = ctypes.c_void_p()
    cfg = N.default_config()
    assert L.nik_create(None, 480, 640, 1, 1, 0, ctypes.byref(ctx)) == N.NIK_ERR_INVALID_ARG
    assert L.nik_create(ctypes.byref(cfg), 481, 640, 1, 1, 0, ctypes.byref(ctx)) == N.NIK_ERR_UNSUPPORTED_SIZE
    # (480 x 642 is a valid geometry since round 4: it runs the any-size kernel family; lengths beyond 8192 are refused)
    assert L.nik_create(ctypes.byref(cfg), 480, 16384, 1, 1, 0, ctypes.byref(ctx)) == N.NIK_ERR_UNSUPPORTED_SIZE
    assert L.nik_create(ctypes.byref(cfg), 480, 640, 0, 1, 0, ctypes.byref(ctx)) == N.NIK_ERR_INVALID_ARG
    assert b"even" in L.nik_last_error(None) or b"positive" in L.nik_last_error(None)
    assert L.nik_synchronize(None) == N.NIK_ERR_INVALID_ARG


def _build_adaptor_test(lib_path, tmpdir):
    import subprocess
    exe = os.path.join(str(tmpdir), "adaptor_test")
    ora = os.path.join(ROOT, "oracle")
    from oracle import kcc_oracle
    kcc_oracle.build()                                       # the test's checker (libkcc_oracle.so)
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "adaptor_test.cpp"),
                           "-I", PKG, "-I", ora, "-L", PKG, "-L", ora, "-Wl,-rpath," + PKG, "-Wl,-rpath," + ora,
                           "-lnislam_kcc_hip", "-lkcc_oracle", "-o", exe])
    return exe


def test_cpp_adaptor_compiles_without_eigen(lib_path, tmp_path):
    """the CorrelationFlow drop-in (ni-slam_amd/correlation_flow_hip.h) compiles with g++ against the C ABI"""
    exe = _build_adaptor_test(lib_path, tmp_path)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_cpp_adaptor_runs(lib_path, tmp_path):
    import subprocess
    exe = _build_adaptor_test(lib_path, tmp_path)
    for size in (["60", "80"], ["480", "640"]):
        out = subprocess.run([exe] + size, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "ADAPTOR TEST OK" in out.stdout, out.stdout + out.stderr


RELEASE_ENV = {"NIK_STREAMS", "NIK_KZZ_CACHE", "NIK_GENERIC", "NIK_GRAPH", "NIK_GROUP_INIT_TIMEOUT", "NIK_GROUP_FORCE_RCCL"}


def _env_names(path):
    data = open(path, "rb").read()
    return set(m.decode() for m in re.findall(rb"(?<![A-Z_0-9])NIK_[A-Z][A-Z_0-9]+(?=\x00)", data)) - {"NIK_OK"}


def test_release_library_carries_no_laboratory_switches(lib_path):
    """the release library reads only the switches a caller needs; ablation flags, LDS padding, the ring-form
    B kernels and the alternative fusion / ordering forms exist in the tuning library (-DKCC_ABLATE) alone -- in the release
    binary not even their names (kcc_tune.h: tune_env() folds to nullptr)."""
    names = {n for n in _env_names(lib_path) if not n.startswith("NIK_ERR")}
    assert names <= RELEASE_ENV, "laboratory switches in the release library: %s" % sorted(names - RELEASE_ENV)
    src = "".join(open(os.path.join(PKG, "csrc", f), errors="ignore").read() for f in os.listdir(os.path.join(PKG, "csrc"))
                  if f.endswith((".hip", ".cpp", ".h")))
    plain = set(re.findall(r'(?<![a-z_])getenv\("(NIK_[A-Z_0-9]+)"\)', src))
    assert plain <= RELEASE_ENV | {"NIK_ABLATE"} and "kBr" not in os.popen("nm -C %s | grep ' kcc::kBr' | head -1" % lib_path).read()


def test_tuning_library_has_the_same_abi(lib_path):
    build = load_module("nislam_build", os.path.join(PKG, "build.py"))
    tune = build.build_tuning()
    T = ctypes.CDLL(tune)
    for n in _declared():
        assert hasattr(T, n), "tuning library lacks %s" % n
    assert {"NIK_ABLATE", "NIK_RING", "NIK_LANE_CUS", "NIK_FUSE_FIX_ZERO"} <= _env_names(tune)
