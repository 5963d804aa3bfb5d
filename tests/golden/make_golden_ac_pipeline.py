"""Golden vectors for area-clearing's non-physics pipeline, produced by the reference's AreaClearingEnv code (run ONLY in the build
container, after `make -C oracle`):

    PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_ac_pipeline.py

Absent third-party primitives are supplied by this repository's restatements (cv2.fillPoly, spfa.spfa, skimage disk; shapely's
Polygon.contains / intersects for convex polygons and the boundary goal points), scipy's are the real ones.  Everything around them
is the reference's own code: update_configuration_space, create_global_shortest_path_to_goal_points, update_global_overhead_map,
generate_observation, boxes_completed, obs_to_goal_difference.  Scene states come from the oracle.  Outputs are data only.
"""
import hashlib
import json
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
from scipy import ndimage

from benchpush_amd import area_clearing_scenario as A
from benchpush_amd.config import default_cfg
from oracle import oracle_bd as ob

for m in ["skimage", "skimage.draw.draw", "pymunk", "pymunk.pygame_util", "pygame", "pynput", "dubins"]:
    sys.modules[m] = MagicMock()


def _orient(p):
    return 1.0 if sum(p[i][0] * p[(i + 1) % len(p)][1] - p[(i + 1) % len(p)][0] * p[i][1] for i in range(len(p))) > 0 else -1.0


class Point:
    def __init__(self, x, y=None):
        self.x, self.y = (x, y) if y is not None else (x[0], x[1])


class Polygon:   # convex polygons only (all shipped layouts): contains = strictly inside, intersects = no separating edge
    def __init__(self, verts):
        self.v = [(float(a), float(b)) for a, b in np.asarray(verts)]

    def contains(self, pt):
        o = _orient(self.v)
        return all(((self.v[(i + 1) % len(self.v)][0] - self.v[i][0]) * (pt.y - self.v[i][1]) - (self.v[(i + 1) % len(self.v)][1] - self.v[i][1]) * (pt.x - self.v[i][0])) * o > 0
                   for i in range(len(self.v)))

    @staticmethod
    def _sep(a, b):
        o = _orient(a)
        for i in range(len(a)):
            j = (i + 1) % len(a)
            if all(((a[j][0] - a[i][0]) * (q[1] - a[i][1]) - (a[j][1] - a[i][1]) * (q[0] - a[i][0])) * o < 0 for q in b):
                return True
        return False

    def intersects(self, other):
        return not (self._sep(self.v, other.v) or self._sep(other.v, self.v))


shp = types.ModuleType("shapely"); shg = types.ModuleType("shapely.geometry")
shg.Polygon, shg.Point, shg.LineString = Polygon, Point, MagicMock()
for name in ("LinearRing", "MultiPolygon", "MultiLineString", "box"):
    setattr(shg, name, MagicMock())
shp.ops = MagicMock(); sys.modules["shapely.ops"] = shp.ops; shp.affinity = MagicMock(); sys.modules["shapely.affinity"] = shp.affinity
shp.geometry = shg


def _fillPoly(img, pts_list, color):
    for pts in pts_list:
        ob.fill_poly(img, [(int(p[0]), int(p[1])) for p in pts], color)
    return img


def _spfa(cmap, source):
    dist, par, _ = ob.spfa(np.asarray(cmap, np.float32), (int(source[0]), int(source[1])))
    return dist, par


def _disk(r):
    r = int(r)
    a = np.arange(-r, r + 1)
    X, Y = np.meshgrid(a, a)
    return (X ** 2 + Y ** 2 <= r ** 2).astype(np.uint8)


cv2 = types.ModuleType("cv2"); cv2.fillPoly = _fillPoly; cv2.line = MagicMock()
spfa = types.ModuleType("spfa"); spfa.spfa = _spfa
skd = types.ModuleType("skimage.draw"); skd.line = lambda r0, c0, r1, c1: ob.sk_line(int(r0), int(c0), int(r1), int(c1)); skd.polygon = MagicMock()
skm = types.ModuleType("skimage.measure"); skm.approximate_polygon = lambda coords, tolerance: ob.approx_polygon(np.asarray(coords), tolerance); skm.block_reduce = MagicMock()
skmo = types.ModuleType("skimage.morphology"); skmo.disk = _disk; skmo.binary_dilation = lambda img, selem: ndimage.binary_dilation(img, structure=selem)
sys.modules.update({"cv2": cv2, "spfa": spfa, "skimage.draw": skd, "skimage.measure": skm, "skimage.morphology": skmo, "shapely": shp, "shapely.geometry": shg})
gym = types.ModuleType("gymnasium"); gym.Env = type("Env", (), {})
spaces = types.ModuleType("gymnasium.spaces"); spaces.Box = lambda *a, **k: None; gym.spaces = spaces
reg = types.ModuleType("gymnasium.envs.registration"); reg.register = lambda **k: None
envs_mod = types.ModuleType("gymnasium.envs"); envs_mod.registration = reg
sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.envs": envs_mod, "gymnasium.envs.registration": reg})

import benchpush.common.evaluation.metrics as refm  # noqa: E402
from benchpush.environments.area_clearing.area_clearing import AreaClearingEnv  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


class _Vec(tuple):
    x = property(lambda s: s[0])
    y = property(lambda s: s[1])


class _Poly:
    def __init__(self, world_verts, position=(0.0, 0.0), angle=0.0):
        self._v = [_Vec((float(x), float(y))) for x, y in world_verts]
        self.body = types.SimpleNamespace(position=_Vec((float(position[0]), float(position[1]))), angle=float(angle), local_to_world=lambda v: v)

    def get_vertices(self):
        return self._v


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


out = {"cases": []}
arrays = {}
for ci, (layout, nsteps, seed) in enumerate([("clear_env", 4, 1), ("walled_env_with_columns", 3, 2), ("clear_env_small", 6, 3)]):
    cfg = default_cfg("area_clearing")
    cfg.env = layout
    trial = A.generate_trials(cfg, 2)[1]
    gp = A.goal_points(cfg)
    o = ob.OracleAreaClearing(A.area_clearing_physics_params(cfg), A.area_clearing_params(cfg), cfg)
    o.reset(trial, observe=False)
    rng = np.random.RandomState(seed)
    actions = [float(a) for a in rng.uniform(-1, 1, nsteps)]
    prev_wv = o.world_verts()
    infos = []
    for a in actions:
        prev_wv = o.world_verts()
        infos.append(o.step(a, observe=False)[4])
    st, wv = o.shape_states(), o.world_verts()
    nbox = len(trial["boxes"])
    AreaClearingEnv._compute_boundary_goals = lambda self, interpolated_points=10: ([], [Point(x, y) for x, y in gp])
    env = AreaClearingEnv(cfg={"render": {"show": False}, "env": layout})
    nst = len(trial["statics"][1])
    nwall = nst - len(env.static_obstacles)
    env.wall_shapes = [_Poly(wv[6 + nbox + k]) for k in range(nwall)]
    env.static_obs_shapes = [_Poly(wv[6 + nbox + k]) for k in range(nwall, nst)]
    env.box_shapes = [_Poly(wv[6 + k], position=st[6 + k, :2], angle=st[6 + k, 2]) for k in range(nbox)]
    # agent: footprint_vertices go through body.local_to_world -> give the mock the real rigid transform
    ca, sa = np.cos(st[0, 2]), np.sin(st[0, 2])
    env.agent = types.SimpleNamespace(body=types.SimpleNamespace(position=_Vec((float(st[0, 0]), float(st[0, 1]))), angle=float(st[0, 2]),
                                                                 local_to_world=lambda v: _Vec((ca * v[0] - sa * v[1] + st[0, 0], sa * v[0] + ca * v[1] + st[0, 1]))))
    env.num_box = nbox
    env.box_clearance_statuses = [False] * nbox
    env.update_configuration_space()
    env.global_overhead_map = env.create_padded_room_zeros()
    env.goal_point_global_map = env.create_global_shortest_path_to_goal_points()
    updated = [np.array(w) for w in wv[6:6 + nbox]]
    num_completed, all_done = env.boxes_completed(updated, env.boundary_polygon, env.box_clearance_statuses)
    obs = env.generate_observation()
    diff = float(refm.obs_to_goal_difference([np.array(w) for w in prev_wv[6:6 + nbox]], updated, env.goal_points, env.boundary_polygon))
    arrays["obs%d" % ci] = obs
    out["cases"].append({"layout": layout, "actions": actions, "cspace_sha": sha(env.configuration_space.astype(np.float32)),
                         "edt_sha": sha(np.asarray(env.closest_cspace_indices).astype(np.int32)), "small_sha": sha(env.small_obstacle_map.astype(np.float32)),
                         "goal_map_sha": sha(env.goal_point_global_map.astype(np.float32)), "overhead_sha": sha(env.global_overhead_map.astype(np.float32)),
                         "num_completed": int(num_completed), "statuses": [bool(s) for s in env.box_clearance_statuses], "last_diff_reward": diff,
                         "oracle_last_info": infos[-1]})
np.savez_compressed(os.path.join(HERE, "ac_pipeline_golden.npz"), **arrays)
with open(os.path.join(HERE, "ac_pipeline_golden.json"), "w") as f:
    json.dump(out, f)
print("wrote ac_pipeline_golden.json / .npz:", [(c["layout"], c["num_completed"], round(c["last_diff_reward"], 4)) for c in out["cases"]])
