"""BoxDeliveryMetric (reference: benchpush/common/metrics/box_pushing_metric.py:5-58): effort score from the cumulative robot
distance and the cumulative box ("cube") distance; the efficiency score is not defined for this task (NotImplementedError)."""
from .base_metric import BaseMetric


class BoxDeliveryMetric(BaseMetric):
    def __init__(self, alg_name, robot_mass) -> None:
        super().__init__(alg_name=alg_name)
        self.robot_mass = robot_mass
        self.reset(None)

    def compute_efficiency_score(self):
        raise NotImplementedError

    def compute_effort_score(self):
        own = self.robot_mass * self.total_robot_dist
        return own / (own + self.total_box_dist)

    def update(self, info, eps_complete=False):
        self.total_box_dist = info["cumulative_cube_distance"]
        self.total_robot_dist = info["cumulative_distance"]
        self.eps_reward = info["cumulative_reward"]
        if eps_complete:
            self.rewards.append(self.eps_reward)
            self.effort_scores.append(self.compute_effort_score())

    def reset(self, info):
        self.eps_reward = 0
        self.total_box_dist = 0
        self.total_robot_dist = 0
