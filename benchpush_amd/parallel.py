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

    `dist` is torch.distributed (initialised) or None for single-process runs.  An initialised group of ONE rank still goes through the
    collective (that is how tests/test_gpu_rccl.py loads RCCL on a one-GPU box).
    """
    if dist is None or not dist.is_initialized():
        return local.clone()
    world = dist.get_world_size()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def gather_episode_block(rows, counts, dist=None):
    """The per-rank [E/R, 6] episode rows of BatchedShipIceEnv.episode_metrics() (efficiency, effort, reward, success, length,
    total_work of every env's last finished episode) plus the finished-episode counts -> the whole job's ([E, 6], [E]) in global env
    order (ranks own contiguous env ranges, so rank-major order is env order).  One all-gather of a fixed [E/R, 7] float64 block:
    4096 envs per GPU = 229 KB per rank, once per evaluation batch."""
    block = torch.cat([rows.to(torch.float64), counts.to(torch.float64).reshape(-1, 1)], dim=1)
    allb = allgather_episode_metrics(block, dist)
    return allb[:, :-1].contiguous(), allb[:, -1].to(torch.int64)


def gather_episode_sums(sums, counts, dist=None):
    """The per-rank [E/R, 6] sums over ALL finished episodes of every env (BatchedShipIceEnv.episode_history()) plus the episode counts -> the whole
    job's ([E, 6], [E]) in global env order: one all-gather of a fixed [E/R, 7] float64 block per evaluation, whatever the number of episodes."""
    block = torch.cat([sums.to(torch.float64), counts.to(torch.float64).reshape(-1, 1)], dim=1)
    allb = allgather_episode_metrics(block, dist)
    return allb[:, :-1].contiguous(), allb[:, -1].to(torch.int64)


def summarize_episode_sums(sums, counts):
    """Means over every finished episode of the job -- the averages of BaseMetric's lists (base_metric.py:12-16)."""
    n = int(counts.sum().item())
    if n == 0:
        return {"episodes": 0}
    mean = (sums.to(torch.float64).sum(dim=0) / n).tolist()
    return {"episodes": n, "efficiency": mean[0], "effort": mean[1], "reward": mean[2], "success_rate": mean[3], "length": mean[4], "total_work": mean[5]}


def summarize_episode_block(rows, counts):
    """Means over the envs that have finished at least one episode -- what BaseMetric's lists hold per algorithm
    (base_metric.py:12-16) reduced to scalars: efficiency, effort, reward, success rate, episode length, total work."""
    m = counts > 0
    n = int(m.sum().item())
    if n == 0:
        return {"envs_with_episode": 0, "episodes": 0}
    r = rows[m].to(torch.float64)
    mean = r.mean(dim=0).tolist()
    return {"envs_with_episode": n, "episodes": int(counts.sum().item()), "efficiency": mean[0], "effort": mean[1], "reward": mean[2],
            "success_rate": mean[3], "length": mean[4], "total_work": mean[5]}
