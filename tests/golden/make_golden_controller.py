"""Golden vectors for the controller side of box-delivery / area-clearing, produced by the reference's own classes (run ONLY in the
build container):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_controller.py

Third-party modules that are absent here are stubbed with MagicMock.  ``skimage.draw.line`` gets a one-pixel stand-in so that
``PositionController.shortest_path(check_straight=True)`` takes its straight-line branch on an all-free map (the branch under test is
the numpy arithmetic around it: target position, room bounds, waypoint headings, the backing-up rule).  Outputs are data only.
"""
import json
import os
import sys
from unittest.mock import MagicMock

import numpy as np

for m in ["shapely", "shapely.geometry", "skimage", "skimage.draw", "skimage.measure", "skimage.draw.draw", "cv2", "pymunk", "gymnasium", "spfa"]:
    sys.modules[m] = MagicMock()
sys.modules["skimage.draw"].line = lambda r0, c0, r1, c1: (np.array([0]), np.array([0]))

from benchpush.common.controller.dp import DP  # noqa: E402
from benchpush.common.controller.position_controller import PositionController  # noqa: E402
from benchpush.common.evaluation.metrics import obs_to_goal_difference, path_length  # noqa: E402
from benchpush.common.metrics.box_pushing_metric import BoxDeliveryMetric  # noqa: E402
from benchpush.common.utils.utils import DotDict  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/benchpush"
rs = np.random.RandomState(11)

# 1. DP.ideal_control / get_setpoint over a pose sequence (box-delivery: defaults; area-clearing: cfg.controller incl. Lfc 0.5)
ac_ctrl = DotDict.load_from_file(os.path.join(REF, "environments/area_clearing/config.yaml")).controller
dp_cases = []
for case in range(10):
    area = case % 2 == 1
    kw = dict(ac_ctrl) if area else dict(dt=0.2, target_speed=0.3)
    p0 = rs.uniform(-3, 3, 2)
    p1 = p0 + rs.uniform(-2, 2, 2)
    pose = np.array([p0[0], p0[1], rs.uniform(-3.1, 3.1)])
    if case >= 6:   # start away from waypoint 0 (what happens after a waypoint switch)
        pose[:2] += rs.uniform(-0.7, 0.7, 2)
    path = np.array([[p0[0], p0[1], 0.0], [p1[0], p1[1], 0.5]])
    dp = DP(x=pose[0], y=pose[1], yaw=pose[2], cx=path.T[0][0:2], cy=path.T[1][0:2], ch=path.T[2][0:2], **kw)
    poses, outs = [], []
    for k in range(60):
        omega, v = dp.ideal_control(pose[0], pose[1], pose[2])
        sp = dp.get_setpoint()
        dp.setpoint = np.asarray(sp)
        poses.append(pose.tolist())
        outs.append([float(omega), float(v[0]), float(v[1]), float(sp[0]), float(sp[1])])
        # any deterministic pose update will do: the test replays the recorded poses
        pose = pose + np.array([v[0] * 2 * 0.02, v[1] * 2 * 0.02, omega * 0.05 * 0.02])
    dp_cases.append({"wp": [p0[0], p0[1], p1[0], p1[1]], "lfc": float(kw.get("Lfc") or 0.0), "target_speed": float(kw["target_speed"]), "dt": float(kw["dt"]),
                     "poses": poses, "out": outs})

# 2. PositionController.get_waypoints_to_spatial_action on an all-free map (straight-line branch)
pc_cases = []
for case in range(12):
    lp, lw = 224, (10.0 if case % 2 == 0 else 24.0)
    ppm = lp / lw
    map_w, map_h = (5.0, 10.0) if case % 2 == 0 else (16.0, 16.0)
    radius = 0.659 if case % 2 == 0 else 1.661
    free = np.ones((600, 600), np.float32)
    idx = np.indices((600, 600))
    pc = PositionController(None, radius, map_w, map_h, free, free, idx, lp, lw, ppm, np.radians(15), 0.05, 0.6, np.radians(10))
    pos = [float(rs.uniform(-map_h / 2 + 0.5, map_h / 2 - 0.5)), float(rs.uniform(-map_w / 2 + 0.5, map_w / 2 - 0.5))]
    heading = float(rs.uniform(-np.pi, np.pi))
    action = int(rs.randint(0, lp * lp))
    path, sign = pc.get_waypoints_to_spatial_action(pos, heading, action)
    pc_cases.append({"lp": lp, "lw": lw, "map_w": map_w, "map_h": map_h, "radius": radius, "pos": pos, "heading": heading, "action": action,
                     "path": [[float(r[0]), float(r[1]), None if r[2] is None else float(r[2])] for r in path], "move_sign": float(sign)})

# 3. path_length, obs_to_goal_difference (no boundary polygon: shapely is stubbed)
class _P:
    def __init__(self, x, y):
        self.x, self.y = x, y
pl = rs.uniform(-5, 5, (7, 2))
goals = [(float(x), float(y)) for x, y in rs.uniform(-5, 5, (6, 2))]
boxes_a = [rs.uniform(0.5, 4, 2) + np.array([[0.5, 0.5], [-0.5, 0.5], [-0.5, -0.5], [0.5, -0.5]]) for _ in range(4)]
boxes_b = [b + rs.uniform(-0.3, 0.3, 2) for b in boxes_a]
misc = {"path": pl.tolist(), "path_length": float(path_length(pl)), "path_cumsum": path_length(pl, cumsum=True).tolist(),
        "goals": goals, "boxes_a": [b.tolist() for b in boxes_a], "boxes_b": [b.tolist() for b in boxes_b],
        "goal_diff": float(obs_to_goal_difference(boxes_a, boxes_b, [_P(*g) for g in goals], None))}

# 4. BoxDeliveryMetric
bm = BoxDeliveryMetric(alg_name="x", robot_mass=1)
bm.reset({})
infos = [{"cumulative_cube_distance": 0.4 * k, "cumulative_distance": 1.1 * k + 0.3, "cumulative_reward": 0.25 * k - 1} for k in range(1, 5)]
for k, i in enumerate(infos):
    bm.update(i, eps_complete=(k == 3))
metric = {"infos": infos, "rewards": bm.rewards, "effort": bm.effort_scores}

# 5. configs
def jsonable(d):
    if isinstance(d, dict):
        return {k: jsonable(v) for k, v in d.items()}
    if isinstance(d, (list, tuple)):
        return [jsonable(v) for v in d]
    return d
cfgs = {"area_clearing": jsonable(DotDict.to_dict(DotDict.load_from_file(os.path.join(REF, "environments/area_clearing/config.yaml"))))}
for name in ("clear_env", "clear_env_small", "walled_env", "walled_env_with_columns"):
    cfgs["area_clearing_env_" + name] = jsonable(DotDict.to_dict(DotDict.load_from_file(os.path.join(REF, "environments/area_clearing/envs/%s.yaml" % name))))

with open(os.path.join(HERE, "controller_golden.json"), "w") as f:
    json.dump({"dp": dp_cases, "position_controller": pc_cases, "misc": misc, "box_delivery_metric": metric, "configs": cfgs}, f)
print("wrote controller_golden.json:", len(dp_cases), "dp cases,", len(pc_cases), "position-controller cases")
