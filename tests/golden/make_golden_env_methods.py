"""Golden vectors from the reference's BoxDeliveryEnv / AreaClearingEnv *classes* (run ONLY in the build container):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_env_methods.py

pymunk, gymnasium, shapely, skimage, cv2 and spfa are absent here; they are stubbed so that the classes can be constructed, and only
methods that never touch them are executed: the episode generators (get_random_robot_start / generate_boundary / generate_boxes on the
env's own RandomState), robot_state_channel, get_local_map (real scipy rotate) and -- with a stand-in "space" that integrates the
kinematic robot body alone (position += velocity * dt, angle += angular_velocity * dt, which is all Chipmunk does to a kinematic body
that touches nothing) -- execute_robot_path with the real DP controller.  Outputs are data only.
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


class _Env:
    pass


class _Box:
    def __init__(self, low=None, high=None, shape=None, dtype=None):
        self.low, self.high, self.shape, self.dtype = low, high, shape, dtype


spaces = types.ModuleType("gymnasium.spaces")
spaces.Box = _Box
gym.Env, gym.spaces = _Env, spaces
reg = types.ModuleType("gymnasium.envs.registration")
reg.register = lambda **k: None
envs_mod = types.ModuleType("gymnasium.envs")
envs_mod.registration = reg
sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.envs": envs_mod, "gymnasium.envs.registration": reg})

from benchpush.environments.area_clearing.area_clearing import AreaClearingEnv  # noqa: E402
from benchpush.environments.box_delivery.box_delivery_env import BoxDeliveryEnv  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}

# 1. episode generators, several consecutive resets per obstacle config ------------------------------------------------
scen = {}
for oc in ["small_empty", "small_columns", "large_columns", "large_divider"]:
    env = BoxDeliveryEnv(cfg={"render": {"show": False, "show_obs": False}, "env": {"obstacle_config": oc}})
    eps = []
    for _ in range(3):   # what init_box_delivery_env draws, in its order (box_delivery_env.py:241-252)
        env.receptacle_position, env.receptacle_size = env.get_receptacle_position_and_size()
        env.robot_info["start_pos"] = env.get_random_robot_start()
        env.boundary_dicts = env.generate_boundary()
        boxes = env.generate_boxes()
        eps.append({"start": [float(v) for v in env.robot_info["start_pos"]],
                    "boundary": [{"type": b["type"], "position": [float(v) for v in b["position"]],
                                  "vertices": np.asarray(b["vertices"]).tolist() if "vertices" in b else None,
                                  "heading": float(b["heading"]) if "heading" in b else None} for b in env.boundary_dicts],
                    "boxes": [[float(b["position"][0]), float(b["position"][1]), float(b["heading"])] for b in boxes]})
    scen[oc] = eps
out["box_delivery_episodes"] = scen
env = BoxDeliveryEnv(cfg={"render": {"show": False, "show_obs": False}})
out["robot_state_channel_rows"] = np.packbits(env.robot_state_channel.astype(np.uint8), axis=1).tolist()
out["padded_room_shape"] = list(env.create_padded_room_zeros().shape)
out["robot_radius"] = float(env.robot_radius)

# 2. get_local_map / get_local_distance_map on a reproducible random image ----------------------------------------------
H, W = out["padded_room_shape"]
gm = (np.random.RandomState(5).randint(0, 9, (H, W)) / 8).astype(np.float32)
local = []
for k, (x, y, h) in enumerate([(0.3, -1.2, 0.7), (-4.1, 2.0, -2.9), (4.4, 1.75, 12.3), (0.0, 0.0, np.pi / 2)]):
    lm = env.get_local_map(gm, (x, y), h)
    ld = env.get_local_distance_map(gm.copy(), (x, y), h)
    local.append({"pose": [x, y, h], "map_u8": (lm * 8).astype(np.uint8), "dist_u8": (ld * 8).astype(np.uint8)})
np.savez_compressed(os.path.join(HERE, "env_methods_golden.npz"), **{"local%d_%s" % (k, key): c[key] for k, c in enumerate(local) for key in ("map_u8", "dist_u8")})
out["local_map_poses"] = [c["pose"] for c in local]


# 3. execute_robot_path in free space ---------------------------------------------------------------------------------------
class _Vec(list):
    @property
    def x(self):
        return self[0]

    @property
    def y(self):
        return self[1]


class _Body:
    def __init__(self, x, y, a):
        self.position, self.angle, self.velocity, self.angular_velocity = _Vec([x, y]), a, [0.0, 0.0], 0.0


class _Space:
    def __init__(self, body):
        self.body = body

    def step(self, dt):   # cpBodyUpdatePosition for a kinematic body without contacts
        b = self.body
        b.position = _Vec([b.position[0] + (b.velocity[0] + 0.0) * dt, b.position[1] + (b.velocity[1] + 0.0) * dt])
        b.angle = b.angle + (b.angular_velocity + 0.0) * dt


def run_path(env, body_attr, start, waypoints, move_sign=1):
    body = _Body(*start)
    holder = types.SimpleNamespace(body=body)
    setattr(env, body_attr, holder)
    env.space = _Space(body)
    env.robot_hit_obstacle = False
    env.dp = None
    env.steps = env.cfg.sim.steps
    env.dt = env.cfg.controller.dt
    if not hasattr(env, "target_speed") or body_attr == "robot":
        env.target_speed = env.cfg.controller.target_speed
    n0 = {"n": 0}
    orig = env.space.step

    def counted(dt):
        n0["n"] += 1
        orig(dt)
    env.space.step = counted
    path = np.array([[w[0], w[1], w[2]] for w in waypoints], dtype=object)
    path[0][2] = None
    env.path = path
    h0 = float(np.mod(start[2] + np.pi, 2 * np.pi) - np.pi)
    dist, turn = env.execute_robot_path([start[0], start[1]], h0, move_sign)
    return {"start": list(start), "waypoints": [[float(w[0]), float(w[1]), float(w[2])] for w in waypoints], "robot_distance": float(dist),
            "turn_angle": float(turn), "final": [float(body.position[0]), float(body.position[1]), float(body.angle)], "sim_steps": n0["n"]}


def headings(pts):
    hs = [0.0]
    for i in range(1, len(pts)):
        a = np.arctan2(pts[i][1] - pts[i - 1][1], pts[i][0] - pts[i - 1][0])
        hs.append(float(np.mod(a + np.pi, 2 * np.pi) - np.pi))
    return hs


rs = np.random.RandomState(21)
paths = []
bd = BoxDeliveryEnv(cfg={"render": {"show": False, "show_obs": False}, "agent": {"action_type": "heading"}})
AreaClearingEnv._compute_boundary_goals = lambda self, interpolated_points=10: ([], [])   # shapely (stubbed) is only used there
ac = AreaClearingEnv(cfg={"render": {"show": False}})
ac.update_global_overhead_map = lambda: None   # periodic raster refresh inside the loop (cv2): no effect on the motion
for case in range(8):
    env_, attr = (bd, "robot") if case % 2 == 0 else (ac, "agent")
    start = (float(rs.uniform(-2, 2)), float(rs.uniform(-1.5, 1.5)), float(rs.uniform(-3, 3)))
    npts = 2 if case < 4 else 3
    pts = [[start[0], start[1]]]
    for _ in range(npts - 1):
        ang, d = rs.uniform(-np.pi, np.pi), rs.uniform(1.0, 1.8)
        pts.append([pts[-1][0] + d * np.cos(ang), pts[-1][1] + d * np.sin(ang)])
    hs = headings(pts)
    wps = [[p[0], p[1], h] for p, h in zip(pts, hs)]
    r = run_path(env_, attr, start, wps)
    r["task"] = "box_delivery" if case % 2 == 0 else "area_clearing"
    paths.append(r)
out["execute_robot_path"] = paths

with open(os.path.join(HERE, "env_methods_golden.json"), "w") as f:
    json.dump(out, f)
print("wrote env_methods_golden.json / .npz;", [(p["task"], p["sim_steps"], round(p["robot_distance"], 4)) for p in paths])
