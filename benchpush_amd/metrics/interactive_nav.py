"""Shared arithmetic of the interactive-navigation scores (Interactive Gibson style) used by ship-ice and maze-NAMO.

Both reference metrics (benchpush/common/metrics/ship_ice_metric.py:26-69, maze_namo_metric.py:25-75) track, per episode,
the summed reward, the agent's path length l0 integrated from the *rounded* ``info['state']``, the work done on the obstacles
(``info['total_work']`` = sum m_i * l_i) and the success flag; they differ only in how the obstacle-free path length L is
obtained at reset.  Scores: efficiency = 1[success] * L / l0, effort = m0*l0 / (m0*l0 + total_work).
"""
import numpy as np

from .base_metric import BaseMetric


class PathEffortMetric(BaseMetric):
    def __init__(self, alg_name, agent_mass):
        super().__init__(alg_name=alg_name)
        self._m0 = agent_mass
        self._clear()

    def _clear(self):
        self.eps_reward = 0
        self._work = 0            # total_mass_dist in the reference
        self._l0 = 0              # total_ship_dist / total_robot_dist
        self.trial_success = False

    # -- reference surface -------------------------------------------------------------------------------------
    def compute_efficiency_score(self):
        return self.L / self._l0 if self.trial_success else 0

    def compute_effort_score(self):
        own = self._m0 * self._l0
        return own / (own + self._work)

    def update(self, info, reward, eps_complete=False):
        self.eps_reward += reward
        self._work = info["total_work"]
        self.trial_success = info["trial_success"]
        here = info["state"]
        self._l0 += np.linalg.norm(np.array(self._state[:2]) - np.array(here[:2]))
        self._state = here
        if eps_complete:
            self.rewards.append(self.eps_reward)
            self.efficiency_scores.append(self.compute_efficiency_score())
            self.effort_scores.append(self.compute_effort_score())

    def reset(self, info):
        self._clear()
        self._state = info["state"]
        self.L = self._free_path_length(info)

    def _free_path_length(self, info):
        raise NotImplementedError
