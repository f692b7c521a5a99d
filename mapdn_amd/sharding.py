"""Multi-GPU: the env batch shards embarrassingly over ranks (one process per GPU).

There is no data-path collective: envs share only read-only constants, and the RNG is keyed by the
GLOBAL env id, so results do not depend on the number of GPUs.  The only communication is the
end-of-rollout gather of per-env episode statistics (RCCL all_gather over xGMI when the backend is
"nccl"; "gloo" on CPU for tests).  The reference has nothing comparable: its "multi-GPU" is shell
scripts pinning independent processes (train_case141.sh:7-21).
"""
from __future__ import annotations

import torch


def shard_range(total_envs: int, world_size: int, rank: int):
    """[lo, hi) of global env ids owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(int(total_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def make_sharded_env(net, profiles, args, total_envs: int, rank: int, world_size: int, device=None, **kw):
    """This rank's slice of a `total_envs`-env batch; env_id_offset keeps the RNG keyed globally."""
    from .env import VoltageControlBatch
    lo, hi = shard_range(total_envs, world_size, rank)
    return VoltageControlBatch(net, profiles, args, n_envs=hi - lo, device=device, env_id_offset=lo, **kw)


def gather_rollout(x: torch.Tensor, sizes=None, force: bool = False):
    """End-of-rollout gather: concatenates each rank's [B_local, ...] tensor along dim 0 on every
    rank, in global env-id order.  `sizes` = per-rank B_local when shards are uneven.  `force`: run the
    collective on a one-rank group too (bench.py --force-dist: first contact with RCCL on one GPU)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return x
    world = dist.get_world_size()
    x = x.contiguous()
    if sizes is None or len(set(sizes)) == 1:
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x)       # one flat collective: payload is KBs, latency-bound
        return out
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)


def reduce_mean_info(info: torch.Tensor, total_envs: int):
    """Mean of the [B_local, 11] info block over ALL envs of the job (one all_reduce of 11 doubles)."""
    import torch.distributed as dist
    s = info.sum(dim=0)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return s / float(total_envs)
