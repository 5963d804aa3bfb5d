"""ShipIceMetric (reference: benchpush/common/metrics/ship_ice_metric.py:5-72): L = goal line - start y.

``BatchedShipIceMetric`` keeps the same quantities as device tensors for E envs and produces the fixed-shape block that
crosses GPUs (benchpush_amd.parallel.allgather_episode_metrics).
"""
import numpy as np
import torch

from .interactive_nav import PathEffortMetric


class ShipIceMetric(PathEffortMetric):
    def __init__(self, alg_name, ship_mass, goal) -> None:
        super().__init__(alg_name, ship_mass)
        self.ship_mass = ship_mass
        self.goal_line = goal[1]

    # names the reference exposes
    @property
    def total_ship_dist(self):
        return self._l0

    @property
    def total_mass_dist(self):
        return self._work

    @property
    def ship_state(self):
        return self._state

    def _free_path_length(self, info):
        return self.goal_line - info["state"][1]


def _round2(t):
    """python round(x, 2) on a float64 tensor (ship_ice_env.py:337-339 rounds info['state']); exact via numpy."""
    return torch.from_numpy(np.array([round(float(v), 2) for v in t.cpu().numpy().ravel()], np.float64).reshape(tuple(t.shape))).to(t.device)


class BatchedShipIceMetric:
    """Device-side accumulators for E envs: episode reward, rounded-state path length, success, total work."""

    def __init__(self, num_envs, ship_mass, goal, device):
        self.ship_mass = float(ship_mass)
        self.goal_line = float(goal[1])
        z = lambda: torch.zeros(num_envs, dtype=torch.float64, device=device)
        self.eps_reward, self.ship_dist, self.total_work, self.success, self.L = z(), z(), z(), z(), z()
        self.prev_xy = torch.zeros((num_envs, 2), dtype=torch.float64, device=device)

    def reset(self, info, mask=None):
        m = torch.ones_like(self.eps_reward, dtype=torch.bool) if mask is None else mask.bool()
        xy = _round2(info[:, 0:2])
        self.prev_xy[m] = xy[m]
        for t in (self.eps_reward, self.ship_dist, self.total_work, self.success):
            t[m] = 0.0
        self.L[m] = self.goal_line - xy[m, 1]

    def update(self, info, reward):
        xy = _round2(info[:, 0:2])
        self.eps_reward += reward
        self.ship_dist += torch.linalg.norm(self.prev_xy - xy, dim=1)
        self.prev_xy = xy
        self.total_work = info[:, 3].clone()
        self.success = info[:, 8].clone()

    def episode_block(self):
        """[E, 6] float64: efficiency, effort, reward, success, path length, total_work (rows valid at episode end)."""
        eff = torch.where(self.success > 0, self.L / self.ship_dist.clamp_min(1e-300), torch.zeros_like(self.L))
        denom = self.ship_mass * self.ship_dist + self.total_work
        effort = torch.where(denom > 0, self.ship_mass * self.ship_dist / denom.clamp_min(1e-300), torch.zeros_like(denom))
        return torch.stack([eff, effort, self.eps_reward, self.success, self.ship_dist, self.total_work], dim=1)
