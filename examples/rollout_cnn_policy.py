"""On-device rollout: uint8 observations go from the environment kernels straight into a torch CNN on the same GPU, actions come
back as device tensors -- no host round trip (the reference trains SB3 PPO("CnnPolicy") on one CPU env,
baselines/ship_ice_nav/ppo/policy.py:29-69).

    python examples/rollout_cnn_policy.py [ship-ice|maze|box|area] [num_envs] [steps]
"""
import sys
import time

import torch
import torch.nn as nn


class SmallCnn(nn.Module):
    def __init__(self, channels, channels_last, action_dim):
        super().__init__()
        self.channels_last = channels_last
        self.net = nn.Sequential(nn.Conv2d(channels, 16, 8, 4), nn.ReLU(), nn.Conv2d(16, 32, 4, 2), nn.ReLU(), nn.AdaptiveAvgPool2d(4), nn.Flatten(),
                                 nn.Linear(32 * 16, 64), nn.ReLU(), nn.Linear(64, action_dim), nn.Tanh())

    def forward(self, obs_u8):
        x = obs_u8.permute(0, 3, 1, 2) if self.channels_last else obs_u8
        return self.net(x.float() / 255.0)


def make_env(kind, n):
    if kind == "ship-ice":
        from benchpush_amd.envs.ship_ice import BatchedShipIceEnv
        return BatchedShipIceEnv(n, cfg={"concentration": 0.3}, num_trials=16), False, 1
    if kind == "maze":
        from benchpush_amd.envs.maze_namo import BatchedMazeEnv
        return BatchedMazeEnv(n, cfg={"num_obstacles": 20}, num_layouts=16), False, 1
    if kind == "box":
        from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
        return BatchedBoxDeliveryEnv(n, num_trials=16), True, 1
    from benchpush_amd.envs.area_clearing import BatchedAreaClearingEnv
    return BatchedAreaClearingEnv(n, num_trials=16), True, 1


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "ship-ice"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    env, channels_last, adim = make_env(kind, n)
    policy = SmallCnn(4, channels_last, adim).to(env.device)
    obs, _ = env.reset()
    ret = torch.zeros(n, dtype=torch.float64, device=env.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(steps):
            act = policy(obs).squeeze(-1).double()          # device tensor in [-1, 1]
            obs, rew, term, trunc, info = env.step(act)
            ret += rew
            env.reset(term | trunc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s: %d envs x %d steps with a CNN policy on the GPU: %.0f env-steps/s, mean return so far %.3f" % (kind, n, steps, n * steps / dt, ret.mean().item()))
    env.check_errors()


if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    main()
