"""Golden vectors of generate_observation_low_dim of the reference's MazeNAMO / BoxDeliveryEnv / AreaClearingEnv / ShipIceEnv classes and
their low_dim_state observation spaces (run ONLY in the build container):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_low_dim.py

Absent third-party modules are stubbed so that the classes import and construct; the methods executed touch numpy only.  Data only.
"""
import json
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

for m in ["shapely", "shapely.geometry", "skimage", "skimage.draw", "skimage.measure", "skimage.morphology", "skimage.draw.draw", "cv2", "pymunk",
          "pymunk.pygame_util", "pygame", "spfa", "pynput", "dubins"]:
    sys.modules[m] = MagicMock()
gym = types.ModuleType("gymnasium")
gym.Env = type("Env", (), {})


class _Box:
    def __init__(self, low=None, high=None, shape=None, dtype=None):
        self.low, self.high, self.shape, self.dtype = low, high, shape, dtype


spaces = types.ModuleType("gymnasium.spaces")
spaces.Box = _Box
gym.spaces = spaces
reg = types.ModuleType("gymnasium.envs.registration")
reg.register = lambda **k: None
envs_mod = types.ModuleType("gymnasium.envs")
envs_mod.registration = reg
sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.envs": envs_mod, "gymnasium.envs.registration": reg})

import benchpush.environments.ship_ice_nav.ship_ice_env as sie  # noqa: E402
from benchpush.environments.area_clearing.area_clearing import AreaClearingEnv  # noqa: E402
from benchpush.environments.box_delivery.box_delivery_env import BoxDeliveryEnv  # noqa: E402
from benchpush.environments.maze_NAMO.maze_NAMO_env import MazeNAMO  # noqa: E402

fake = {"exp": {c: {k: {"goal": (0, 9), "ship_state": (6, 1, np.pi / 2), "obstacles": []} for k in range(8)} for c in (0.1, 0.2, 0.3, 0.4, 0.5)}}
sie.open = lambda *a, **k: None
sie.pickle = types.SimpleNamespace(load=lambda f: fake)

rng = np.random.RandomState(12)
polys = []
for k in range(6):   # convex polygons of 4..9 vertices anywhere in a +-8 m square (centroids of both signs)
    n = 4 + k
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    r = rng.uniform(0.3, 1.2)
    c = rng.uniform(-8, 8, 2)
    polys.append((np.stack([np.cos(ang), np.sin(ang)], 1) * r + c).tolist())
P = [np.asarray(p) for p in polys]
out = {"polys": polys, "robot": [3.25, -1.5]}
me = types.SimpleNamespace(robot_body=types.SimpleNamespace(position=types.SimpleNamespace(x=3.25, y=-1.5)))
out["maze"] = MazeNAMO.generate_observation_low_dim(me, P).tolist()
out["box_delivery"] = BoxDeliveryEnv.generate_observation_low_dim(None, P).tolist()
out["area_clearing"] = AreaClearingEnv.generate_observation_low_dim(None, P).tolist()
out["ship_ice"] = sie.ShipIceEnv.generate_observation_low_dim(None, P).tolist()
sp = {}
e = MazeNAMO(cfg={"low_dim_state": True, "num_obstacles": 7}); sp["maze_random"] = [list(e.observation_space.shape), str(np.dtype(e.observation_space.dtype))]
e = MazeNAMO(cfg={"low_dim_state": True, "randomize_obstacles": False}); sp["maze_fixed"] = [list(e.observation_space.shape), str(np.dtype(e.observation_space.dtype))]
e = BoxDeliveryEnv(cfg={"low_dim_state": True, "render": {"show": False, "show_obs": False}, "boxes": {"num_boxes_small": 7}})
sp["box_delivery"] = [list(e.observation_space.shape), str(np.dtype(e.observation_space.dtype))]
AreaClearingEnv._compute_boundary_goals = lambda self, interpolated_points=10: ([], [])   # shapely only
e = AreaClearingEnv(cfg={"low_dim_state": True, "render": {"show": False, "show_obs": False}})
sp["area_clearing"] = [list(e.observation_space.shape), str(np.dtype(e.observation_space.dtype)), int(e.cfg.num_obstacles)]
out["spaces"] = sp
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "low_dim_golden.json"), "w") as f:
    json.dump(out, f)
print({k: (v if k == "spaces" else len(v)) for k, v in out.items()})
