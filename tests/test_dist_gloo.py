"""world_size-2 gloo tests (CPU) of the multi-GPU rules the product applies: nik_group_shard (which rank owns which pair /
candidate) and nik_group_pick_best (the reference's winner rule over the gathered records, loop_closure.cc:61-65) are the
library's own host functions, called through the C ABI; the two exchanges (4-double all-reduce, 8-double all-gather) run
over torch.distributed/gloo here and over RCCL inside kcc_group.cpp on GPUs.  The per-rank compute engine in this file is the
CPU oracle (the GPU path is covered by the -m gpu tests); the check is sharded == unsharded."""
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

from kcc_helpers import SMALL, nik  # noqa: E402


def residual_stats(results):
    """[sum PSR_t, sum PSR_r, sum |t|^2, count] -- what k_residual_stats reduces on the device (kcc_kernels.hip)"""
    s = [0.0, 0.0, 0.0, 0.0]
    for r in results:
        s[0] += r["info"][0]; s[1] += r["info"][2]; s[2] += r["pose"][0] ** 2 + r["pose"][1] ** 2; s[3] += 1.0
    return torch.tensor(s, dtype=torch.float64)


def record_of(global_index, result):
    """the 8-double record nik_group_gather_best exchanges: [score, global index, pose x3, info x3]"""
    if global_index < 0:
        return torch.tensor([0.0, -1.0, 0, 0, 0, 0, 0, 0], dtype=torch.float64)
    return torch.tensor([sum(result["info"]), float(global_index), *result["pose"], *result["info"]], dtype=torch.float64)


def test_shard_partitions():
    G = nik().Group
    for n in (0, 1, 7, 32, 255, 256, 4096):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                b, e = G.shard(n, world, r)
                assert 0 <= b <= e <= n and (e - b) in (n // world, n // world + 1)
                cover += list(range(b, e))
            assert cover == list(range(n))


def test_pick_best_rules():
    G = nik().Group
    rec = lambda s, i: [s, i, 0, 0, 0, 0, 0, 0]                      # noqa: E731
    assert G.pick_best([rec(25.0, 6)]) == 0
    assert G.pick_best([rec(0.0, -1)]) == -1                          # no candidate on the only rank
    assert G.pick_best([rec(-6.0, 0)]) == -1                          # below the initial (-1,-1,-1) response: never selected
    assert G.pick_best([rec(10.0, 3), rec(10.0, 9)]) == 0             # strict '>': the first in global order keeps a tie
    assert G.pick_best([rec(10.0, 3), rec(10.5, 9), rec(10.5, 20)]) == 1
    assert G.pick_best([rec(0.0, -1), rec(1.0, 4)]) == 1


def _worker(rank, world, port, n_pairs, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synth
    from oracle import kcc_oracle as ko
    G = nik().Group
    g = SMALL
    cfg = ko.default_config(rotation_divisor=g["PD"], rotation_channel=g["PC"])
    keys, curs, _ = synth.make_batch(n_pairs, g["H"], g["W"], seed0=40)        # every rank regenerates the same data
    b, e = G.shard(n_pairs, world, rank)
    poses, infos, _, _ = ko.track_pairs(cfg, keys[b:e], curs[b:e], True)
    stats = residual_stats([dict(pose=list(p), info=list(i)) for p, i in zip(poses, infos)])
    dist.all_reduce(stats)
    # loop closure: candidates = the keys, query = curs[0]; each rank scores its shard
    orc = ko.Oracle(cfg, g["H"], g["W"])
    q = orc.normalize_u8(curs[0])
    _, qp = orc.intermedium(q)
    best, best_score, best_res = -1, -3.0, None
    for i in range(b, e):
        kf, kp = orc.intermedium(orc.normalize_u8(keys[i]))
        pose, info, _ = orc.compute_pose(kf, q, kp, qp, False)
        if info.sum() > best_score:
            best, best_score, best_res = i, info.sum(), dict(pose=list(pose), info=list(info))
    mine = record_of(best, best_res)
    allrec = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allrec, mine)
    allrec = torch.stack(allrec).numpy()
    win = G.pick_best(allrec)
    out_q.put((rank, stats.tolist(), int(allrec[win][1]) if win >= 0 else -1, allrec[win].tolist() if win >= 0 else None))
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
    want = residual_stats([dict(pose=list(p), info=list(i)) for p, i in zip(poses, infos)]).tolist()
    for rank, stats, gi, rec in got:
        assert stats == pytest.approx(want, rel=1e-12)
        assert stats[3] == n_pairs
    assert got[0][2] == got[1][2] == 0 and got[0][3] == got[1][3]      # curs[0] matches keys[0]; both ranks agree
    assert (got[0][3][2], got[0][3][3]) == (poses[0][0], poses[0][1])
