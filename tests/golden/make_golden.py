"""Generate golden vectors from the importable parts of the reference (run ONLY in the build container).

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

Third-party modules that are absent here (pymunk, skimage, cv2, shapely, gymnasium) are stubbed with MagicMock; only
reference functions that never touch them are executed.  Outputs are data (inputs + expected outputs) written next
to this script; no reference source is copied.
"""
import json
import os
import random
import sys
from unittest.mock import MagicMock

import numpy as np

for m in ["shapely", "shapely.geometry", "skimage", "skimage.draw", "skimage.measure", "skimage.draw.draw", "cv2",
          "pymunk", "gymnasium"]:
    sys.modules[m] = MagicMock()

from benchpush.common.evaluation.metrics import euclid_dist, total_work_done  # noqa: E402
from benchpush.common.geometry.polygon import generate_polygon, poly_area, poly_centroid  # noqa: E402
from benchpush.common.metrics.maze_namo_metric import MazeNamoMetric  # noqa: E402
from benchpush.common.metrics.ship_ice_metric import ShipIceMetric  # noqa: E402
from benchpush.common.occupancy_grid import occupancy_map as om  # noqa: E402
from benchpush.common.utils.utils import DotDict  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/benchpush"

# 1. polygons -----------------------------------------------------------------------------------------
polys = []
for seed in range(12):
    random.seed(seed)
    d = 0.8 + 0.1 * seed
    origin = (1.0 + seed, 3.0 + 2 * seed)
    p = generate_polygon(d, origin)
    polys.append({"seed": seed, "diameter": d, "origin": origin, "vertices": p.tolist(),
                  "area": float(poly_area(p)), "centroid": [float(c) for c in poly_centroid(p)]})
# a polygon straddling the axes exercises the abs() in poly_centroid
neg = np.array([[-1.0, -2.0], [3.0, -1.5], [2.5, 2.0], [-0.5, 1.0]])
polys.append({"seed": None, "vertices": neg.tolist(), "area": float(poly_area(neg)),
              "centroid": [float(c) for c in poly_centroid(neg)]})

# 2. work ------------------------------------------------------------------------------------------------
work_cases = []
rs = np.random.RandomState(7)
for case in range(6):
    random.seed(100 + case)
    a_list, b_list = [], []
    for k in range(5 + case):
        p = generate_polygon(1.0 + 0.1 * k, (2.0 + k, 5.0 + 0.5 * k))
        th = rs.uniform(-0.2, 0.2)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        c = p.mean(axis=0)
        q = (p - c) @ R.T + c + rs.uniform(-0.3, 0.3, size=2)
        a_list.append(p)
        b_list.append(q)
    work_cases.append({"a": [p.tolist() for p in a_list], "b": [p.tolist() for p in b_list],
                       "work": float(total_work_done(a_list, b_list))})
work_cases.append({"a": [neg.tolist()], "b": [(neg + np.array([0.1, 0.05])).tolist()],
                   "work": float(total_work_done([neg], [neg + np.array([0.1, 0.05])]))})

# 3. occupancy-grid crops (numpy-only methods) ----------------------------------------------------------
og = om.OccupancyGrid(1 / 25, 1 / 25, 12, 40, 6, 6, None, 25)
dims = {"occ_map_width": og.occ_map_width, "occ_map_height": og.occ_map_height,
        "local_window_height": og.local_window_height, "local_window_width": og.local_window_width}
edt_cases = []
for goal_y, state in [(9, (6.0, 1.0, np.pi / 2)), (9, (0.3, 8.7, 1.0)), (19, (11.9, 17.3, 2.0)), (9, (5.5, 39.0, 1.2)),
                      (9, (-0.2, 3.33, 0.4))]:
    e = og.ego_view_goal_dist_transform(goal_y, state, 2)
    edt_cases.append({"goal_y": goal_y, "state": list(state), "u8": (e * 255).astype(np.uint8)})
# crop arithmetic with an injected global map: global[i, j] = (i * 7 + j * 3) % 256 / 255
gi, gj = np.meshgrid(np.arange(1000), np.arange(300), indexing="ij")
gmap = ((gi * 7 + gj * 3) % 251) / 250.0
om.block_reduce = lambda img, block, fn: img
crop_cases = []
for state in [(6.0, 1.0, np.pi / 2), (0.05, 0.5, 1.0), (11.99, 38.9, 2.0), (3.777, 20.123, 0.1)]:
    c = og.ego_view_obstacle_map(gmap, state, 2)
    og._compute_global_footprint = lambda ship_state, ship_vertices, padding=0.25: gmap
    f = og.ego_view_footprint(state, None, 2)
    og.global_orientation_map = lambda ship_state, head, tail: gmap
    o = og.ego_view_orientation_map(state, None, None, 2)
    assert np.array_equal(c, f) and np.array_equal(c, o)
    crop_cases.append({"state": list(state), "u8": (c * 255).astype(np.uint8)})
np.savez_compressed(os.path.join(HERE, "occupancy_golden.npz"),
                    edt=np.stack([c["u8"] for c in edt_cases]),
                    edt_goal=np.array([c["goal_y"] for c in edt_cases], np.float64),
                    edt_state=np.array([c["state"] for c in edt_cases], np.float64),
                    crop=np.stack([c["u8"] for c in crop_cases]),
                    crop_state=np.array([c["state"] for c in crop_cases], np.float64))

# 4. metrics ---------------------------------------------------------------------------------------------
metric_cases = []
rs = np.random.RandomState(3)
for ep in range(4):
    m = ShipIceMetric("alg", ship_mass=1, goal=(0, 9))
    x, y = 6.0, 1.0
    infos = [{"state": (round(x, 2), round(y, 2), round(np.pi / 2, 2)), "total_work": 0.0}]
    m.reset(infos[0])
    tw = 0.0
    steps = []
    n = 30 + 3 * ep
    for t in range(n):
        x += rs.uniform(-0.05, 0.05)
        y += 0.24
        tw += max(0.0, rs.uniform(-0.1, 0.2))
        done = t == n - 1
        info = {"state": (round(x, 2), round(y, 2), 1.57), "total_work": tw, "trial_success": bool(done and ep % 2 == 0)}
        r = float(rs.uniform(-2, 1))
        m.update(info, r, done)
        steps.append({"info": {"state": list(info["state"]), "total_work": tw, "trial_success": info["trial_success"]},
                      "reward": r, "done": done})
    metric_cases.append({"reset_info": {"state": list(infos[0]["state"]), "total_work": 0.0}, "steps": steps,
                         "efficiency": m.efficiency_scores, "effort": m.effort_scores, "rewards": m.rewards})

# 4b. maze metric ------------------------------------------------------------------------------------------
maze_metric_cases = []
rs = np.random.RandomState(5)
goal_dt = (np.arange(240 * 240, dtype=np.float64).reshape(240, 240) % 977) + 1.0
for ep in range(3):
    m = MazeNamoMetric("alg", robot_mass=1)
    x, y = 11.25, 3.75
    info0 = {"state": (round(x, 2), round(y, 2), 1.57), "total_work": 0.0, "goal_dt": goal_dt, "m_to_pix_scale": 16}
    m.reset(info0)
    tw = 0.0
    steps = []
    n = 20 + 5 * ep
    for t in range(n):
        x += rs.uniform(-0.03, 0.03)
        y += 0.12
        tw += max(0.0, rs.uniform(-0.05, 0.1))
        done = t == n - 1
        info = {"state": (round(x, 2), round(y, 2), 1.57), "total_work": tw, "trial_success": bool(done and ep != 1)}
        r = float(rs.uniform(-2, 1))
        m.update(info, r, done)
        steps.append({"info": {"state": list(info["state"]), "total_work": tw, "trial_success": info["trial_success"]}, "reward": r, "done": done})
    maze_metric_cases.append({"reset_state": list(info0["state"]), "steps": steps, "efficiency": m.efficiency_scores,
                              "effort": m.effort_scores, "rewards": m.rewards})

# 5. configs -----------------------------------------------------------------------------------------------
def jsonable(d):
    if isinstance(d, dict):
        return {k: jsonable(v) for k, v in d.items()}
    if isinstance(d, (list, tuple)):
        return [jsonable(v) for v in d]
    return d


cfgs = {name: jsonable(DotDict.to_dict(DotDict.load_from_file(os.path.join(REF, "environments", name, "config.yaml"))))
        for name in ["ship_ice_nav", "maze_NAMO", "box_delivery"]}

with open(os.path.join(HERE, "reference_golden.json"), "w") as f:
    json.dump({"polygons": polys, "work": work_cases, "grid_dims": dims, "metrics": metric_cases, "maze_metrics": maze_metric_cases, "configs": cfgs,
               "euclid": float(euclid_dist((1.0, 2.0), (4.0, 6.0)))}, f)
print("wrote golden fixtures")
