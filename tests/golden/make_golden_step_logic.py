"""Golden vectors for ShipIceEnv.step's control / constraint / reward / termination logic, produced by the reference class itself
(run ONLY in the build container):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_step_logic.py

pymunk and friends are absent: the class is constructed with them stubbed (the missing ice-field pickle is replaced by an in-memory
dict), and step() runs on a stand-in space that integrates the kinematic ship alone (position += velocity * dt, angle +=
angular_velocity * dt) with one static floe far away, so that the 400-sub-step loop, the yaw / boundary rules, the reward terms,
termination and the info dict come from the reference's code.  Outputs are data only.
"""
import json
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

for m in ["shapely", "shapely.geometry", "skimage", "skimage.draw", "skimage.measure", "skimage.morphology", "skimage.draw.draw", "cv2",
          "pymunk.pygame_util", "pygame", "spfa", "pynput", "dubins"]:
    sys.modules[m] = MagicMock()


class Vec2d(tuple):
    def __new__(cls, x, y):
        return tuple.__new__(cls, (x, y))
    x = property(lambda s: s[0])
    y = property(lambda s: s[1])


pm = MagicMock()
pm.Vec2d = Vec2d
sys.modules["pymunk"] = pm
gym = types.ModuleType("gymnasium")
gym.Env = type("Env", (), {})
spaces = types.ModuleType("gymnasium.spaces")
spaces.Box = lambda *a, **k: None
gym.spaces = spaces
reg = types.ModuleType("gymnasium.envs.registration")
reg.register = lambda **k: None
envs_mod = types.ModuleType("gymnasium.envs")
envs_mod.registration = reg
sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.envs": envs_mod, "gymnasium.envs.registration": reg})

import benchpush.environments.ship_ice_nav.ship_ice_env as sie  # noqa: E402

FLOE = np.array([[6.5, 30.5], [5.5, 30.5], [5.5, 29.5], [6.5, 29.5]])
fake = {"exp": {c: {0: {"goal": (0, 9), "ship_state": (6, 1, np.pi / 2), "obstacles": []}} for c in (0.1, 0.2, 0.3, 0.4, 0.5)}}
sie.open = lambda *a, **k: None          # the ice-field pickle is one of the repository's missing large blobs
sie.pickle = types.SimpleNamespace(load=lambda f: fake)


class _Body:
    def __init__(self, x, y, a):
        self.position, self.angle, self.velocity, self.angular_velocity = Vec2d(x, y), a, Vec2d(0.0, 0.0), 0.0


class _Space:
    def __init__(self, body):
        self.body = body

    def step(self, dt):   # cpBodyUpdatePosition of a kinematic body that touches nothing
        b = self.body
        b.position = Vec2d(b.position[0] + (b.velocity[0] + 0.0) * dt, b.position[1] + (b.velocity[1] + 0.0) * dt)
        b.angle = b.angle + (b.angular_velocity + 0.0) * dt


class _Poly:
    def __init__(self, verts):
        c = verts.mean(0)
        self._v = [tuple(p) for p in (verts - c)]
        self.body = types.SimpleNamespace(angle=0.0, position=np.array(c))

    def get_vertices(self):
        return self._v


def run(start, actions):
    env = sie.ShipIceEnv()
    env.steps, env.dt = env.cfg.sim.steps, env.cfg.dt
    env.total_work = [0, []]
    env.ship_body = _Body(*start)
    env.space = _Space(env.ship_body)
    env.polygons = [_Poly(FLOE)]
    env.prev_obs = sie.CostMap.get_obs_from_poly(env.polygons)
    env.t = 0
    env.generate_observation = lambda: None
    rows = []
    for a in actions:
        _, r, term, trunc, info = env.step(a)
        b = env.ship_body
        rows.append({"action": float(a), "reward": float(r), "terminated": bool(term), "truncated": bool(trunc), "pose": [float(b.position[0]), float(b.position[1]), float(b.angle)],
                     "state": [float(v) for v in info["state"]], "total_work": float(info["total_work"]), "dist_reward": float(info["dist reward"]),
                     "trial_success": bool(info["trial_success"]), "angular_velocity": float(b.angular_velocity)})
        if term:
            break
    return {"start": list(start), "steps": rows}


rs = np.random.RandomState(3)
cases = [run((6.0, 1.0, np.pi / 2), [float(np.float32(a)) for a in rs.uniform(-1, 1, 12)]),
         run((0.15, 5.0, 2.8), [1.0] * 6),               # leaves the channel on the left: -50 and termination
         run((6.0, 8.7, 1.3), [0.2] * 6),                # reaches the goal line: +200
         run((6.0, 2.0, 3.0), [1.0] * 5),                # yaw reaches pi: the turn rate is zeroed for the rest of the step
         run((11.9, 3.0, 0.3), [-0.5] * 6),              # leaves on the right
         run((6.0, 3.0, 0.12), [-1.0] * 5)]              # yaw reaches 0
# ---- MazeNAMO.step (maze_NAMO_env.py:402-485) on the same kind of stand-in space, with an injected goal map -----------------
import benchpush.environments.maze_NAMO.maze_NAMO_env as mz  # noqa: E402


def maze_map(h, w):
    i, j = np.indices((h, w))
    return ((i * 37 + j * 91) % 1000) / 1000.0


def run_maze(start, actions, box_centre=(2.0, 13.0)):
    env = mz.MazeNAMO()
    env.steps, env.dt, env.target_speed = env.cfg.sim.steps, env.cfg.dt, env.cfg.target_speed
    env.total_work = [0, []]
    env.robot_body = _Body(*start)
    env.space = _Space(env.robot_body)
    c = np.array(box_centre)
    env.polygons = [_Poly(c + np.array([[0.5, 0.5], [-0.5, 0.5], [-0.5, -0.5], [0.5, -0.5]]))]
    env.prev_obs = mz.CostMap.get_obs_from_poly(env.polygons)
    env.goal = (env.cfg.env.goal_x, env.cfg.env.goal_y)
    env.wall_collision = False
    env.prev_dist_value = None
    s = env.cfg.occ.m_to_pix_scale
    env.global_distance_map = maze_map(int(env.cfg.env.length * s), int(env.cfg.env.width * s))
    env.t = 0
    env.generate_observation = lambda: None
    rows = []
    for a in actions:
        _, r, term, trunc, info = env.step(a)
        b = env.robot_body
        rows.append({"action": float(a), "reward": float(r), "terminated": bool(term), "pose": [float(b.position[0]), float(b.position[1]), float(b.angle)],
                     "dist_increment": float(info["dist increment reward"]), "trial_success": bool(info["trial_success"]), "total_work": float(info["total_work"])})
        if term:
            break
    return {"start": list(start), "box": list(box_centre), "steps": rows}


maze_cases = [run_maze((11.25, 3.75, np.pi / 2), [float(np.float32(a)) for a in rs.uniform(-1, 1, 10)]),
              run_maze((5.8, 3.75, np.pi), [0.0] * 6),                 # drives into the goal radius: +200
              run_maze((3.0, 12.0, -0.4), [0.7] * 8, box_centre=(12.0, 13.0))]
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_logic_golden.json"), "w") as f:
    json.dump({"floe": FLOE.tolist(), "ship_ice": cases, "maze": maze_cases}, f)
print("maze:", [(len(c["steps"]), c["steps"][-1]["terminated"], round(c["steps"][-1]["reward"], 3)) for c in maze_cases])
print("wrote step_logic_golden.json:", [(len(c["steps"]), c["steps"][-1]["terminated"], round(c["steps"][-1]["reward"], 3)) for c in cases])
