"""Multi-GPU sharding helpers.

The env.step() path shards by environment with no exchange: rank r of R owns the contiguous
global env ids [r*E, (r+1)*E) and scenario selection is a function of the global id, so results do not depend on R.
The only collective is the gather of per-episode metrics -- what the reference accumulates in python lists
(benchpush/common/metrics/base_metric.py:12-16, ship_ice_metric.py:57-60) -- done with one all_gather
(backend "nccl" == RCCL over xGMI on ROCm, "gloo" in the CPU tests).
"""
import torch


def shard_range(total_envs, rank, world):
    """Contiguous env-id range [lo, hi) owned by `rank`; the remainder goes to the lowest ranks."""
    base, rem = divmod(int(total_envs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_episode_metrics(local, dist=None):
    """All-gather a fixed-shape [n, k] float64 metric block from every rank -> [world*n, k] (rank-major).

    `dist` is torch.distributed (initialised) or None for single-process runs.
    """
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local.clone()
    world = dist.get_world_size()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out
