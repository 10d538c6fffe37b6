"""nik_group (multi-GPU C ABI): what a 1-GPU box can check.

* nik_group_shard / nik_group_pick_best, the host rules of the group: tests/test_dist_gloo.py (CPU, two gloo ranks).
* a local group of one GPU: sharded tracking / loop closure through the group equal the direct context calls bit for bit,
  and the device-reduced residual statistics equal the host sum over the per-pair results.
* the same through RCCL itself (NIK_GROUP_FORCE_RCCL=1: ncclCommInitAll / ncclCommInitRank with one rank, all-reduce and
  all-gather on the device) -- exercises the dlopen'ed RCCL call path; communicators of more than one GPU need the 8-GPU node.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import synth
from kcc_helpers import SMALL, nik


def _host_stats(res):
    return np.array([sum(r["info"][0] for r in res), sum(r["info"][2] for r in res),
                     sum(r["pose"][0] ** 2 + r["pose"][1] ** 2 for r in res), float(len(res))])


def _group_checks():
    N = nik()
    g = SMALL
    cfg = N.default_config(rotation_divisor=g["PD"], rotation_channel=g["PC"])
    n = 24
    keys, curs, _ = synth.make_batch(n, g["H"], g["W"], seed0=70)
    # direct context
    cf = N.CorrelationFlow(cfg, g["H"], g["W"], max_batch=n, max_frames=2 * n + 2)
    for i in range(n):
        cf.intermedium_u8(keys[i], i)
    import torch
    dc = torch.from_numpy(curs).cuda()
    torch.cuda.synchronize()
    cf.set_residual_stats(True)
    want = [r.as_dict() for r in cf.track_batch_dev(dc.data_ptr(), list(range(n)), list(range(n, 2 * n)), True, sync=True)]
    st = cf.residual_stats()
    assert np.allclose(st, _host_stats(want), rtol=1e-12, atol=0) and st[3] == n
    want_lr = cf.pose_batch(list(range(n)), list(range(n, 2 * n)), False)
    assert np.allclose(cf.residual_stats(), _host_stats(want_lr), rtol=1e-12, atol=0)
    cf.intermedium_u8(curs[3], 2 * n)
    bi, _, bres = cf.match(2 * n, list(range(n)))
    # local group of one GPU
    grp = N.Group.local(cfg, g["H"], g["W"], max_batch=n, max_frames=2 * n + 2, devices=[0])
    assert grp.world == 1
    f0 = grp.flows[0]
    for i in range(n):
        f0.intermedium_u8(keys[i], i)
    got = grp.track_batch(curs, list(range(n)), list(range(n, 2 * n)), True)
    assert got == want
    assert np.allclose(grp.allreduce_residual(), _host_stats(want), rtol=1e-12, atol=0)
    grp.allreduce_residual(wait=False)
    assert np.allclose(grp.residual_result(), _host_stats(want), rtol=1e-12, atol=0)
    bm, bl, gres = grp.match(curs[3], 2 * n, [list(range(n))])
    assert (bm, bl) == (0, bi) and gres == bres
    assert grp.gather_best([-1], [N.NikPoseResult()])[0] == -1
    grp.close()
    # one-process-per-GPU form, world of one
    grp2 = N.Group.rank(cf, 0, 1)
    cf.track_batch_dev(dc.data_ptr(), list(range(n)), list(range(n, 2 * n)), True, sync=True)
    assert np.allclose(grp2.allreduce_residual(), _host_stats(want), rtol=1e-12, atol=0)
    grp2.close()
    cf.close()


@pytest.mark.gpu
def test_local_group_of_one_gpu_equals_direct_calls():
    _group_checks()


@pytest.mark.gpu
def test_group_through_rccl_single_rank():
    """the same checks with the collectives going through RCCL (own process: the switch is read at group creation)"""
    env = dict(os.environ, NIK_GROUP_FORCE_RCCL="1")
    code = ("import sys; sys.path[:0] = [%r, %r]; import test_group; test_group._group_checks(); "
            "from kcc_helpers import nik; print('RCCL-LIB', *nik().Group.rccl_library()); print('RCCL-OK')") % (
        os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL-OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
    # which RCCL the library bound is a reported fact (bench.py's multi_gpu object): a path naming librccl, and whether it is the
    # copy the process had loaded already (the helpers import torch, whose wheel carries its own librccl: then RTLD_NOLOAD hits)
    lib = [l for l in p.stdout.splitlines() if l.startswith("RCCL-LIB")][0].split()
    assert "rccl" in lib[1] and lib[2] in ("True", "False"), lib


@pytest.mark.gpu
def test_comm_init_deadline_turns_a_hang_into_an_error():
    """a rank whose peers never arrive: nik_group_create_rank(world = 2) on its own returns an error after
    $NIK_GROUP_INIT_TIMEOUT seconds instead of blocking in ncclCommInitRank for good (on the 8-GPU box a hang becomes a
    `fallback: true` line, not a driver timeout).  A 2-rank RCCL communicator cannot FORM on one device -- that leg of
    configs[3] stays for the 8-GPU node; what is tested here is the failure path."""
    env = dict(os.environ, NIK_GROUP_INIT_TIMEOUT="4")
    code = ("import sys, time; sys.path[:0] = [%r, %r]; from kcc_helpers import SMALL, nik; N = nik()\n"
            "cf = N.CorrelationFlow(N.default_config(rotation_divisor=SMALL['PD'], rotation_channel=SMALL['PC']), SMALL['H'], SMALL['W'], max_batch=4, max_frames=4)\n"
            "t0 = time.time()\n"
            "try:\n    N.Group.rank(cf, 0, 2, N.Group.unique_id()); print('FORMED')\n"
            "except Exception as e:\n    print('DEADLINE %%.1f' %% (time.time() - t0), str(e)[:200])\n"
            "sys.stdout.flush(); import os; os._exit(0)\n") % (
        os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "DEADLINE" in p.stdout and "did not return within 4 s" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def _run_bench(world, gb, dump, tmp_path, workload="pairs", bare=False):
    """bare: the literal `python bench.py --gpus N` (bench.py starts its own ranks); else under torch.distributed.run, the
    form the driver uses when it launches the ranks itself"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NIK_BENCH_GLOBAL_BATCH=str(gb), NIK_BENCH_DUMP=str(dump), NIK_BENCH_DEVICE="0", NIK_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    common = ["bench.py", "--gpus", str(world), "--workload", workload, "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--no-profile", "--no-cached",
              "--no-live-prof", "--repeats", "2"]
    if world == 1 or bare:
        cmd = [sys.executable] + common
    else:
        port = 29600 + (os.getpid() % 1000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + common
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    import json
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    return line, [json.load(open("%s.%d" % (dump, r))) for r in range(world)]


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_two_ranks_on_one_device_equal_one_rank(tmp_path):
    """the HIP path run as two ranks (each its contiguous shard of ONE global batch; both on device 0, gloo for the
    rendezvous) gives bit for bit the per-pair results of the unsharded run, and the all-reduced statistics agree"""
    gb = 48
    line1, d1 = _run_bench(1, gb, tmp_path / "w1", tmp_path)
    line2, d2 = _run_bench(2, gb, tmp_path / "w2", tmp_path, bare=True)        # `python bench.py --gpus 2`, nothing around it
    assert line2["n_gpus"] == 2 and line1["n_gpus"] == 1
    assert line2["multi_gpu"]["world"] == 2 and line2["timing"]["regions"] == 2
    whole = d1[0]["results"]
    parts = d2[0]["results"] + d2[1]["results"]
    assert len(whole) == gb == len(parts)
    assert parts == whole
    s1 = np.array(d1[0]["stats"]); s2 = np.array(d2[0]["stats"])
    assert s1[3] == gb == s2[3] and np.allclose(s1, s2, rtol=1e-12, atol=0) and np.array_equal(np.array(d2[1]["stats"]), s2)


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_configs3_hd_two_ranks_equal_one_rank(tmp_path):
    """BASELINE configs[3] as written -- 1280x720 RGB pairs sharded over the ranks with the residual all-reduce -- runs through
    bench.py's multi-rank path (`--workload hd --gpus N`): two ranks on one device give the unsharded run's results bit for
    bit, and the line carries the machine-checkable multi-GPU facts."""
    gb = 12
    line1, d1 = _run_bench(1, gb, tmp_path / "h1", tmp_path, "hd")
    line2, d2 = _run_bench(2, gb, tmp_path / "h2", tmp_path, "hd")
    assert "1280x720" in line2["metric"] and line2["n_gpus"] == 2
    assert d2[0]["results"] + d2[1]["results"] == d1[0]["results"] and len(d1[0]["results"]) == gb
    s1 = np.array(d1[0]["stats"]); s2 = np.array(d2[0]["stats"])
    assert s1[3] == gb == s2[3] and np.allclose(s1, s2, rtol=1e-12, atol=0)
    mg = line2["multi_gpu"]
    assert mg["world"] == 2 and mg["fallback"] is False and mg["rccl_ranks"] == 0        # (gloo test hook: no RCCL communicator)
    assert 0 < mg["pairs_per_s_per_rank_min"] <= mg["pairs_per_s_per_rank_max"]
    assert line1["multi_gpu"]["world"] == 1


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_loop_closure_candidate_shards_two_ranks_equal_one_rank(tmp_path):
    """SURVEY 8(e) loop closure: the CANDIDATE SET sharded over the ranks (contiguous shards, the query on every rank, every
    rank's best record gathered, winner by loop_closure.cc:61-65 with ties to the lowest global index) -- `bench.py --workload
    loop4096 --gpus 2` on one device (gloo rendezvous; a 2-rank RCCL communicator cannot form on one GPU) names the same winner
    with the same score as the unsharded run, and three ranks (uneven shards) do too."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import json

    def run(world):
        dump = tmp_path / ("lc%d" % world)
        env = dict(os.environ, NIK_BENCH_DUMP=str(dump), NIK_BENCH_DEVICE="0", NIK_BENCH_BACKEND="gloo")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
        p = subprocess.run([sys.executable, "bench.py", "--gpus", str(world), "--workload", "loop4096", "--candidates", "200", "--steps", "5", "--warmup", "2"],
                           cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        return line, [json.load(open("%s.%d" % (dump, r))) for r in range(world)]
    l1, d1 = run(1)
    assert l1["parity_spot_check"] is True and l1["winner"]["index"] == 44 and l1["n_gpus"] == 1
    for world in (2, 3):
        lw, dw = run(world)
        assert lw["n_gpus"] == world and lw["scaling"] == "strong" and lw["parity_spot_check"] is True
        assert sum(lw["multi_gpu"]["candidates_per_rank"]) == 200 and lw["multi_gpu"]["world"] == world
        for r in range(world):                                   # every rank sees the same winner: the unsharded run's
            assert dw[r]["best"] == d1[0]["best"] == 44
            assert dw[r]["result"]["pose"] == d1[0]["result"]["pose"][:3] and dw[r]["result"]["info"] == d1[0]["result"]["info"][:3]


def test_bench_refuses_a_world_that_is_not_gpus():
    """`--gpus N` under a launcher that created another world size is an error, not a line with the wrong n_gpus (CPU: the
    check runs before anything is imported)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2"], cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--workload", "pyramid"], cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "single-GPU" in p.stderr
