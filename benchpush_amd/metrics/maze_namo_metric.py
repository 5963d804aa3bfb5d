"""MazeNamoMetric (reference: benchpush/common/metrics/maze_namo_metric.py:5-75): L is read from the wavefront goal map handed
over in the reset info: ``goal_dt[int(y * s), int(x * s)] / s`` with ``s = m_to_pix_scale``."""
from .interactive_nav import PathEffortMetric


class MazeNamoMetric(PathEffortMetric):
    def __init__(self, alg_name, robot_mass) -> None:
        super().__init__(alg_name, robot_mass)
        self.robot_mass = robot_mass

    @property
    def total_robot_dist(self):
        return self._l0

    @property
    def total_mass_dist(self):
        return self._work

    @property
    def robot_state(self):
        return self._state

    def _free_path_length(self, info):
        s = info["m_to_pix_scale"]
        x, y = info["state"][0], info["state"][1]
        return info["goal_dt"][int(y * s), int(x * s)] / s
