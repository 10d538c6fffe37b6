"""bench.py's accounting helpers (CPU): the contract's bytes per pair, the stream choice, and the roofline object's design /
contract fractions (no kernel may be priced on bytes it does not touch; the 1.10x generosity of the nominal
planes must be a number in the line)."""
import argparse
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_contract_bytes_per_pair(bench):
    # 40.63 MB per 640x480 pair (30.17 with the Kzz cache), 83.08 MB at 1280x720, 47.26 MB per loop-closure candidate
    assert abs(bench.algorithmic_bytes(480, 640, 720, 480) / 1e6 - 40.62) < 0.02
    assert abs(bench.algorithmic_bytes(480, 640, 720, 480, kzz_cached=True) / 1e6 - 30.16) < 0.02
    assert abs(bench.algorithmic_bytes(720, 1280, 720, 480) / 1e6 - 83.06) < 0.03
    assert abs(bench.algorithmic_bytes(480, 640, 720, 480, hypotheses=2, with_intermedium=False) / 1e6 - 47.26) < 0.03


def test_stream_choice(bench, monkeypatch):
    monkeypatch.delenv("NIK_STREAMS", raising=False)
    a = argparse.Namespace(streams=0)
    assert bench._streams(a, 512, pairs=True) == 2 and bench._streams(a, 256, pairs=True) == 3 and bench._streams(a, 512, pairs=False) == 3
    monkeypatch.setenv("NIK_STREAMS", "1")
    assert bench._streams(a, 512, pairs=True) == 1                     # (tools/rocprof_summary.py profiles on one stream)
    assert bench._streams(argparse.Namespace(streams=4), 512, pairs=True) == 4


def test_roofline_fractions(bench):
    B = 10
    contract = 40.0e6
    kernels = [dict(name="kB<480,fwd_mul_inv>", avg_ms=1.0, share=0.6, bytes_per_launch=4.0e9, design_bytes_per_launch=3.6e9, gbps=3600.0),
               dict(name="kA_inv<240,shifted>", avg_ms=0.5, share=0.4, bytes_per_launch=0.4e9 * 1.0, design_bytes_per_launch=0.2e9, gbps=400.0)]
    r = bench._roofline(kernels, B, live=dict(stats={}, traffic={}), contract_bytes_per_unit=contract)
    assert r["kernel"] == "kB<480,fwd_mul_inv>" and abs(r["frac"] - 4.0e9 / 1e-3 / 8e12) < 1e-4
    assert abs(r["frac_design"] - 3.6e9 / 1e-3 / 8e12) < 1e-4 and r["frac_design"] < r["frac"]
    gen = (4.0e9 + 0.4e9) / B / contract
    assert abs(r["nominal_bytes_over_contract"] - gen) < 1e-3 and abs(r["frac_contract"] - r["frac"] / gen) < 1e-3


@pytest.mark.gpu
def test_stage_profiler_reports_design_bytes():
    """nik_profile_read: every stage carries its nominal planes and the bytes it is built to move; the symmetry shortcuts show
    (design < nominal for the column-trimmed and Hermitian-half kernels), the u8 kernel's frame-store copy shows (design > nominal),
    and no stage is free"""
    import numpy as np
    import torch
    import synth
    from kcc_helpers import FULL, nik
    N = nik()
    H, W = FULL["H"], FULL["W"]
    cf = N.CorrelationFlow(N.default_config(), H, W, max_batch=4, max_frames=8)
    keys, curs, _ = synth.make_batch(4, H, W, seed0=3)
    dk = torch.from_numpy(keys).cuda(); dc = torch.from_numpy(curs).cuda(); torch.cuda.synchronize()
    cf.intermedium_batch_dev(dk.data_ptr(), 4, [0, 1, 2, 3])
    cf.set_streams(1); cf.profile_enable(True)
    cf.track_batch_dev(dc.data_ptr(), [0, 1, 2, 3], [4, 5, 6, 7], True, sync=True)
    st = {s["name"]: s for s in cf.profile_read() if s["launches"]}
    cf.profile_enable(False); cf.close()
    assert len(st) >= 13 and all(s["bytes"] > 0 and s["bytes_design"] > 0 and s["ms"] > 0 for s in st.values())
    for n in ("kA_inv<240,shifted>", "kB<640,fwd_abs_inv>", "kB<480,fwd_mul_inv>", "kA_inv<360,kernel_fwd>", "kB<640,solve_inv>"):
        assert st[n]["bytes_design"] < st[n]["bytes"], n
    assert st["kA_fwd<240,u8>"]["bytes_design"] > st["kA_fwd<240,u8>"]["bytes"]
    for n in ("kA_fwd<240,rot8>", "kA_fwd<360,polar>", "kA_inv<240,argmax>"):
        assert st[n]["bytes_design"] == st[n]["bytes"], n
    C = 8.0 * (H // 2 + 1) * W
    assert abs(st["kA_inv<240,argmax>"]["bytes"] / st["kA_inv<240,argmax>"]["launches"] - 4 * C) < 1.0
