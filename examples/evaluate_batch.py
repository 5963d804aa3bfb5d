"""Batched counterpart of BasePolicy.evaluate (benchpush/baselines/ship_ice_nav/ppo/policy.py:73-99): run a policy on E environments until
every env has finished `episodes` episodes and report the benchmark's scores -- efficiency, effort, reward, success -- from the on-device
episode lists (ring of the last episodes + sums over all of them: no per-step host round trip, one check every 50 steps).  Multi-GPU: launch with
torch.distributed.run; every rank owns E envs and the [E/R, 7] block of sums and counts is all-gathered once at the end.

    python examples/evaluate_batch.py [E=1024] [episodes=2] [concentration=0.3]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from benchpush_amd.envs.ship_ice import BatchedShipIceEnv
from benchpush_amd.parallel import gather_episode_sums, summarize_episode_sums


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
    steps = 0
    while True:
        obs, rew, term, trunc, info = env.step(straight_ahead_policy(obs, info))
        age += 1
        done = term.bool() | (age >= 300)                      # TimeLimit of the registered id (environments/__init__.py:6)
        obs, info = env.reset(done)                            # device mask, no host round trip: a reset of a running episode closes it as truncated,
        age = torch.where(done, torch.zeros_like(age), age)    # and the returned obs / info carry the first observation of the new episodes
        steps += 1
        if steps % 50 == 0:                                    # the only host synchronisation: every 50 steps, are we there yet?
            _, _, cnt = env.episode_history()
            if int(cnt.min().item()) >= episodes:
                break
    env.check_errors()
    ring, sums, cnt = env.episode_history()                    # every finished episode is accounted on the device (lists: env.episode_lists())
    alls, allc = gather_episode_sums(sums, cnt, dist)          # one [E/R, 7] all-gather for the whole evaluation
    if rank == 0:
        print("all ranks, every finished episode:", summarize_episode_sums(alls, allc))
        lists = env.episode_lists()
        print("rank 0, env 0: %d episodes, efficiency / effort of the last ones:" % int(cnt[0].item()), [tuple(round(float(v), 3) for v in r[:2]) for r in lists[0][-3:]])
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
