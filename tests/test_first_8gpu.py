"""tools/first_8gpu.py (the scripted first multi-GPU run): its assertions on synthetic bench lines (CPU), and
the script itself on a one-GPU box (--selftest: N = 1 plus two ranks on one device through the gloo hook)."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mod():
    spec = importlib.util.spec_from_file_location("first_8gpu", os.path.join(ROOT, "tools", "first_8gpu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _line(n, ranks=None, fallback=False, lo=100.0, hi=104.0):
    return dict(n_gpus=n, value=lo * n, multi_gpu=dict(world=n, rccl_ranks=n if ranks is None else ranks, fallback=fallback,
                                                       pairs_per_s_per_rank_min=lo, pairs_per_s_per_rank_max=hi))


def test_assertions_catch_what_they_are_for():
    m = _mod()
    assert m.check(_line(8), "pairs", 8) == []
    assert m.check(_line(1, ranks=0), "pairs", 1) == []                       # one GPU: no communicator expected
    assert any("rccl_ranks" in b for b in m.check(_line(8, ranks=1), "pairs", 8))
    assert any("fallback" in b for b in m.check(_line(4, fallback=True), "hd", 4))
    assert any("balance" in b for b in m.check(_line(2, hi=115.0), "pairs", 2))
    assert any("n_gpus" in b for b in m.check(_line(4), "pairs", 8))
    lc = dict(n_gpus=2, value=1.0, parity_spot_check=True, config=dict(candidates=10), multi_gpu=dict(world=2, rccl_ranks=2, fallback=False, candidates_per_rank=[5, 5]))
    assert m.check(lc, "loop4096", 2) == []
    lc["multi_gpu"]["candidates_per_rank"] = [5, 4]
    assert any("add up" in b for b in m.check(lc, "loop4096", 2))
    lc["parity_spot_check"] = False
    assert any("winner" in b for b in m.check(lc, "loop4096", 2))
    # the one-device gloo hook: no RCCL communicator can form, balance is not asserted
    assert m.check(_line(2, ranks=0, hi=150.0), "pairs", 2, selftest_gloo=True) == []


@pytest.mark.gpu
@pytest.mark.timeout(1700)
def test_script_selftest_on_one_gpu(tmp_path):
    out = tmp_path / "first.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "first_8gpu.py"), "--selftest", "--quick", "--out", str(out)],
                       cwd=ROOT, capture_output=True, text=True, timeout=1600)
    assert p.returncode == 0 and "all assertions hold" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
