"""MazeNamoMetric: efficiency / interaction-effort scores of one maze-NAMO episode.

Same arithmetic and call protocol as benchpush/common/metrics/maze_namo_metric.py:5-75: the obstacle-free path length L
is read from the wavefront map handed over in the reset info (``goal_dt[int(y*s), int(x*s)] / s``).
"""
import numpy as np

from .base_metric import BaseMetric


class MazeNamoMetric(BaseMetric):
    def __init__(self, alg_name, robot_mass) -> None:
        super().__init__(alg_name=alg_name)
        self.eps_reward = 0
        self.total_mass_dist = 0
        self.robot_mass = robot_mass
        self.total_robot_dist = 0

    def compute_efficiency_score(self):
        if not self.trial_success:
            return 0
        return self.L / self.total_robot_dist

    def compute_effort_score(self):
        return (self.robot_mass * self.total_robot_dist) / (self.robot_mass * self.total_robot_dist + self.total_mass_dist)

    def update(self, info, reward, eps_complete=False):
        self.eps_reward += reward
        self.total_mass_dist = info["total_work"]
        self.trial_success = info["trial_success"]
        robot_state = info["state"]
        self.total_robot_dist += np.linalg.norm(np.array(self.robot_state[:2]) - np.array(robot_state[:2]))
        self.robot_state = robot_state
        if eps_complete:
            self.rewards.append(self.eps_reward)
            self.efficiency_scores.append(self.compute_efficiency_score())
            self.effort_scores.append(self.compute_effort_score())

    def reset(self, info):
        self.eps_reward = 0
        self.total_mass_dist = 0
        self.total_robot_dist = 0
        self.trial_success = False
        self.robot_state = info["state"]
        goal_dt = info["goal_dt"]
        m_to_pix_scale = info["m_to_pix_scale"]
        robot_pixel_x = int(self.robot_state[0] * m_to_pix_scale)
        robot_pixel_y = int(self.robot_state[1] * m_to_pix_scale)
        self.L = goal_dt[robot_pixel_y, robot_pixel_x] / m_to_pix_scale
