"""world_size-2 gloo tests (CPU) of the multi-GPU glue: sharding of pairs / candidates across ranks and the two
collectives.  The per-rank compute engine here is the CPU oracle (the GPU path is covered by the -m gpu tests);
the check is that sharded == unsharded."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from kcc_helpers import PKG, SMALL, load_module  # noqa: E402

kd = load_module("kcc_dist", os.path.join(PKG, "kcc_dist.py"))


def test_shard_range_partitions():
    for n in (0, 1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                b, e = kd.shard_range(n, world, r)
                assert 0 <= b <= e <= n and (e - b) in (n // world, n // world + 1)
                cover += list(range(b, e))
            assert cover == list(range(n))


def _worker(rank, world, port, n_pairs, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synth
    from oracle import kcc_oracle as ko
    g = SMALL
    cfg = ko.default_config(rotation_divisor=g["PD"], rotation_channel=g["PC"])
    keys, curs, _ = synth.make_batch(n_pairs, g["H"], g["W"], seed0=40)        # every rank regenerates the same data
    b, e = kd.shard_range(n_pairs, world, rank)
    poses, infos, _, _ = ko.track_pairs(cfg, keys[b:e], curs[b:e], True)
    res = [dict(pose=list(p), info=list(i)) for p, i in zip(poses, infos)]
    stats = kd.allreduce_residual_stats(kd.residual_stats(res))
    # loop closure: candidates = the keys, query = curs[0]; each rank scores its shard
    orc = ko.Oracle(cfg, g["H"], g["W"])
    q = orc.normalize_u8(curs[0])
    _, qp = orc.intermedium(q)
    best, best_score, best_res = -1, -3.0, None
    for i in range(b, e):
        kf, kp = orc.intermedium(orc.normalize_u8(keys[i]))
        pose, info, _ = orc.compute_pose(kf, q, kp, qp, False)
        if info.sum() > best_score:
            best, best_score, best_res = i - b, info.sum(), dict(pose=list(pose), info=list(info))
    gi, rec = kd.gather_best_match(best, best_res, b)
    out_q.put((rank, stats.tolist(), gi, rec))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_matches_single_rank():
    import synth
    from oracle import kcc_oracle as ko
    n_pairs, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference
    g = SMALL
    cfg = ko.default_config(rotation_divisor=g["PD"], rotation_channel=g["PC"])
    keys, curs, _ = synth.make_batch(n_pairs, g["H"], g["W"], seed0=40)
    poses, infos, _, _ = ko.track_pairs(cfg, keys, curs, True)
    want = kd.residual_stats([dict(pose=list(p), info=list(i)) for p, i in zip(poses, infos)]).tolist()
    for rank, stats, gi, rec in got:
        assert stats == pytest.approx(want, rel=1e-12)
        assert stats[3] == n_pairs
    assert got[0][2] == got[1][2] == 0 and got[0][3] == got[1][3]      # curs[0] matches keys[0]; both ranks agree
    assert (got[0][3][2], got[0][3][3]) == (poses[0][0], poses[0][1])


def test_gather_best_match_single_process_rules():
    r = dict(pose=[1.0, 2.0, 0.0], info=[10.0, 10.0, 5.0])
    assert kd.gather_best_match(2, r, 4)[0] == 6
    assert kd.gather_best_match(-1, None, 0) == (-1, None)
    low = dict(pose=[0, 0, 0], info=[-2.0, -2.0, -2.0])              # sum below the initial (-1,-1,-1) response: never selected
    assert kd.gather_best_match(0, low, 0) == (-1, None)
