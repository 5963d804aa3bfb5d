"""Batched counterpart of BasePolicy.evaluate (benchpush/baselines/ship_ice_nav/ppo/policy.py:73-99): run a policy on E environments until
every env has finished `episodes` episodes and report the benchmark's scores -- efficiency, effort, reward, success -- from the on-device
episode metrics (no per-step host round trip).  Multi-GPU: launch with torch.distributed.run; every rank owns E envs and the [E/R, 6] episode
rows are all-gathered once at the end.

    python examples/evaluate_batch.py [E=1024] [episodes=2] [concentration=0.3]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from benchpush_amd.envs.ship_ice import BatchedShipIceEnv
from benchpush_amd.parallel import gather_episode_block, summarize_episode_block


def straight_ahead_policy(obs, info):
    """Steer back towards heading pi/2 (a stand-in for model.predict(obs)): action in [-1, 1] from the ship's heading in info[:, 2]."""
    return torch.clamp((torch.pi / 2 - info[:, 2]) * 2.0, -1.0, 1.0)


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    episodes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    conc = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
    rank, world, dist = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    env = BatchedShipIceEnv(E, cfg={"concentration": conc, "random_start": True, "start_x_range": 11}, env_id_offset=rank * E,
                            device="cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")))
    obs, info = env.reset()
    age = torch.zeros(E, dtype=torch.int64, device=env.device)
    sums = torch.zeros(6, dtype=torch.float64, device=env.device)
    nfin = 0
    finished = torch.zeros(E, dtype=torch.int64, device=env.device)
    while int(finished.min().item()) < episodes:
        obs, rew, term, trunc, info = env.step(straight_ahead_policy(obs, info))
        age += 1
        done = term.bool() | (age >= 300)                      # TimeLimit of the registered id (environments/__init__.py:6)
        if bool(done.any()):
            env.reset(done)                                    # a reset of a running episode closes it as truncated
            rows, cnt = env.episode_metrics()                  # rows of the envs that just finished are fresh
            sums += rows[done].sum(dim=0)
            nfin += int(done.sum().item())
            finished += done.to(torch.int64)
            age[done] = 0
    env.check_errors()
    rows, cnt = env.episode_metrics()
    allr, allc = gather_episode_block(rows, cnt, dist)
    if rank == 0:
        mean = (sums / max(nfin, 1)).tolist()
        print("rank 0: %d episodes: efficiency %.3f effort %.3f reward %.1f success %.2f length %.1f total_work %.2f" % (nfin, *mean))
        print("all ranks, last episode of every env:", summarize_episode_block(allr, allc))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
