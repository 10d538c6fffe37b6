"""Multi-GPU glue of the KCC front end (one process per GPU, torch.distributed; backend "nccl" == RCCL over xGMI on
MI355X, "gloo" in the CPU tests).

The path shards by independent units: frame pairs for tracking, candidate keyframes for loop
closure.  No data-path collective is needed; the two collectives below are the only exchanges:
  * allreduce_residual_stats -- one 4-double all-reduce per batch: [sum PSR_t, sum PSR_r, sum |t|^2, count]
    (the "pose-graph residual sum" the north star attaches to the batch);
  * gather_best_match        -- all-gather of each rank's best loop-closure candidate (8 doubles) followed by the
    reference's selection rule (loop_closure.cc:61-65: strictly larger response.sum() wins, so the first
    candidate in global order wins ties).
Nothing here touches the GPU kernels or the oracle; tensors live on whatever device the caller passes.
"""
import torch
import torch.distributed as dist


def shard_range(n, world, rank):
    """contiguous shard [b, e) of n units for `rank` of `world`; sizes differ by at most one, order preserved"""
    base, rem = divmod(n, world)
    b = rank * base + min(rank, rem)
    return b, b + base + (1 if rank < rem else 0)


def residual_stats(results, device="cpu"):
    """[sum PSR_t, sum PSR_r, sum |t|^2, count] of a list of pose results (dicts or NikPoseResult)"""
    s = [0.0, 0.0, 0.0, 0.0]
    for r in results:
        pose, info = (r["pose"], r["info"]) if isinstance(r, dict) else (r.pose, r.info)
        s[0] += info[0]
        s[1] += info[2]
        s[2] += pose[0] ** 2 + pose[1] ** 2
        s[3] += 1.0
    return torch.tensor(s, dtype=torch.float64, device=device)


def allreduce_residual_stats(stats, group=None):
    """in-place sum over ranks (4 doubles: latency-bound, one collective per batch)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats


def gather_best_match(local_best_index, local_results, shard_begin, group=None, device="cpu"):
    """local_best_index: index into this rank's candidate shard (-1 if none); local_results: its pose result.
    Returns (global_index, record) identical on every rank, record = [score, idx, pose x3, info x3]."""
    rec = torch.full((8,), float("-inf"), dtype=torch.float64, device=device)
    rec[1] = -1.0
    if local_best_index >= 0:
        r = local_results
        pose, info = (r["pose"], r["info"]) if isinstance(r, dict) else (list(r.pose), list(r.info))
        rec = torch.tensor([info[0] + info[1] + info[2], float(shard_begin + local_best_index), *pose, *info],
                           dtype=torch.float64, device=device)
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if world > 1:
        out = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(out, rec, group=group)
    else:
        out = [rec]
    best, best_score = None, -3.0                         # LoopClosureResult(): response(-1,-1,-1)
    for t in out:                                         # rank order == global candidate order (contiguous shards)
        if t[1] >= 0 and float(t[0]) > best_score:
            best, best_score = t, float(t[0])
    return (int(best[1]), best.tolist()) if best is not None else (-1, None)
