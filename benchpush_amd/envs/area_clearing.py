"""area-clearing-v0 on MI355X: batched tensor environment + the reference-shaped single-env adapter.

Reference: benchpush/environments/area_clearing/area_clearing.py (AreaClearingEnv).  Same engine and kernels as box-delivery-v0
with ``bp_bd_config.task = 1``: waypoints from the PositionController, execute_robot_path under the DP controller (look-ahead
Lfc 0.5, velocity x5, omega / 2), then ``sim.steps`` more sim steps; a box is cleared when its polygon no longer intersects the
clearance boundary; rewards from the change of each box's distance to the nearest boundary goal point.

Layouts: trial t = ``random.Random(base_seed + t)`` with the reference's draw order (it uses the unseeded global ``random``).
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..area_clearing_scenario import area_clearing_params, area_clearing_physics_params, env_layout, generate_trials, goal_points
from ..config import default_cfg, merge_user_cfg
from ..gym_shim import Env, spaces
from .box_delivery import BatchedBoxDeliveryEnv, low_dim_observation

__all__ = ["BatchedAreaClearingEnv", "AreaClearingEnv", "AC_INFO_KEYS"]
AC_INFO_KEYS = ["x", "y", "theta", "total_work", "collision_reward", "diff_reward", "box_completed_reward", "box_count", "ministeps",
                "robot_hit_obstacle", "substeps", "robot_distance", "t", "num_waypoints", "work", "pushing_reward"]


def _ac_cfg(cfg):
    c = merge_user_cfg(default_cfg("area_clearing"), cfg)
    if c.env not in c.envs:
        raise FileNotFoundError(f"Environment config {c.env} not found")   # area_clearing.py:105-108
    if c.agent.action_type not in ("heading", "position", "velocity"):
        raise ValueError("agent.action_type must be heading, position or velocity")
    return c


class BatchedAreaClearingEnv(BatchedBoxDeliveryEnv):
    """E independent area-clearing environments on one GPU: reset(mask) / step(actions) with device tensors."""

    def __init__(self, num_envs, cfg=None, trials=None, device="cuda:0", env_id_offset=0, num_trials=32, base_seed=0):
        if not torch.cuda.is_available():
            raise _lib.BpError("BatchedAreaClearingEnv needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.L = _lib.load()
        self.cfg = _ac_cfg(cfg)
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self.params = area_clearing_physics_params(self.cfg)
        self.bd_params = area_clearing_params(self.cfg)
        if trials is None:
            trials = generate_trials(self.cfg, num_trials, base_seed)
        self.trials = trials
        nbox = len(trials[0]["boxes"])
        self.nbox = nbox
        self.bd_params["num_boxes"] = nbox
        lay = env_layout(self.cfg)
        self.goal_points = goal_points(self.cfg)
        bcfg = _lib.make_bd_config(self.params, self.bd_params, self.cfg)
        _lib.fill_area_geometry(bcfg, lay.boundary, lay.outer_boundary, self.cfg.agent.footprint_vertices, self.goal_points)
        bcfg.distance_scale_max = self.bd_params["distance_scale_max"]
        self.h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self.L, None, self.L.bp_bd_create(C.byref(bcfg), self.num_envs, int(env_id_offset), dev_index, C.byref(self.h)), "bp_bd_create")
        c = lambda a, dt: np.ascontiguousarray(a, dt)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        starts = c(np.stack([t["start"] for t in trials]), np.float64)
        boxes = c(np.stack([t["boxes"] for t in trials]), np.float64)
        sv = c(np.stack([t["statics"][0] for t in trials]), np.float64)
        sc = c(np.stack([t["statics"][1] for t in trials]), np.int32)
        sp = c(np.stack([t["statics"][2] for t in trials]), np.float64)
        sr = c(np.stack([t["statics"][3] for t in trials]), np.float64)
        st = c(np.stack([t["statics"][4] for t in trials]), np.int32)
        _lib.check(self.L, self.h, self.L.bp_bd_load(self.h, len(trials), nbox, p(starts), p(boxes), sv.shape[1], p(sv), p(sc), p(sp), p(sr), p(st)), "bp_bd_load")
        self._alloc_io()
        lp = self.L.bp_obs_height(self.h)
        self.obs_shape = (lp, lp, 4)
        self.obs = torch.zeros((self.num_envs,) + self.obs_shape, dtype=torch.uint8, device=self.device)
        self.action_dim = 2 if self.cfg.agent.action_type == "velocity" else 1
        self._actions = torch.zeros(self.num_envs * self.action_dim, dtype=torch.float64, device=self.device)


class AreaClearingEnv(Env):
    """Reference-shaped single environment (E = 1): reset()/step() returns and info keys of area_clearing.py:563-778."""

    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 4}

    def __init__(self, cfg=None, trials=None, device="cuda:0", num_trials=32, **kwargs):
        super().__init__()
        self._b = BatchedAreaClearingEnv(1, cfg=cfg, trials=trials, device=device, num_trials=num_trials)
        self.cfg = self._b.cfg
        from ..obs_log import refuse_render_log_obs
        refuse_render_log_obs(self.cfg, "area-clearing-v0")
        lay = env_layout(self.cfg)
        self.boundary_vertices, self.outer_boundary_vertices = lay.boundary, lay.outer_boundary
        self.walls = lay.walls if "walls" in lay else []
        self.static_obstacles = lay.static_obstacles if "static_obstacles" in lay else []
        self.goal_points = [tuple(g) for g in self._b.goal_points]
        self.num_box = self._b.nbox
        self.max_yaw_rate_step = (np.pi / 2) / 15
        lp = self._b.obs_shape[0]
        if self.cfg.agent.action_type == "velocity":      # area_clearing.py:200-205
            self.action_space = spaces.Box(low=-1, high=1, shape=(2,), dtype=np.float32)
        elif self.cfg.agent.action_type == "position":
            self.action_space = spaces.Box(low=0, high=lp * lp, shape=(1,), dtype=np.int32)
        else:
            self.action_space = spaces.Box(low=-1, high=1, shape=(1,), dtype=np.float32)
        self.observation_shape = self._b.obs_shape
        self.low_dim_state = self.cfg.low_dim_state
        if self.low_dim_state:                            # area_clearing.py:212-215
            self.fixed_trial_idx = self.cfg.fixed_trial_idx
            self.observation_space = spaces.Box(low=-10, high=30, shape=(self.cfg.num_obstacles * 2,), dtype=np.float32)
        else:
            self.observation_space = spaces.Box(low=0, high=255, shape=self.observation_shape, dtype=np.uint8)
        self.episode_idx = None
        self.box_clearance_statuses = [False] * self.num_box

    def _boxes(self):
        verts, cnt = self._b.world_polys()
        verts, cnt = verts[0].cpu().numpy(), cnt[0].cpu().numpy()
        return [verts[6 + k, : cnt[6 + k]].copy() for k in range(self.num_box)]

    def _statuses(self, boxes):
        """box_clearance_statuses: the box polygon no longer intersects the (convex) clearance boundary (area_clearing.py:1122-1140)."""
        bd = np.asarray(self.boundary_vertices, np.float64)

        def sep(a, b):
            o = 1.0 if np.sum(a[:, 0] * np.roll(a[:, 1], -1) - np.roll(a[:, 0], -1) * a[:, 1]) > 0 else -1.0
            for i in range(len(a)):
                e = a[(i + 1) % len(a)] - a[i]
                cr = e[0] * (b[:, 1] - a[i, 1]) - e[1] * (b[:, 0] - a[i, 0])
                if np.all(cr * o < 0):
                    return True
            return False
        return [bool(sep(bd, np.asarray(b)) or sep(np.asarray(b), bd)) for b in boxes]

    def _low_dim(self, boxes):
        return low_dim_observation(boxes)

    def _result(self, info):
        """low_dim_state: the vector is the observation; otherwise it rides along in info (area_clearing.py:599-606,766-772)."""
        if self.low_dim_state:
            return info.pop("low_level_observation")
        return self._b.obs[0].cpu().numpy()

    def reset(self, seed=None, options=None):
        self.episode_idx = 0 if self.episode_idx is None else self.episode_idx + 1
        if self.episode_idx:
            self._b.check_errors()   # capacity flags of the episode that just ended (raises BpError)
        self._b.reset()
        it = self._b.info[0].cpu().numpy()
        boxes = self._boxes()
        self.box_clearance_statuses = [False] * self.num_box
        info = {"state": (round(float(it[0]), 2), round(float(it[1]), 2), round(float(it[2]), 2)), "total_work": 0, "obs": boxes, "box_count": 0,
                "boundary": self.boundary_vertices, "walls": self.walls, "static_obstacles": self.static_obstacles,
                "goal_positions": self.goal_points, "low_level_observation": self._low_dim(boxes)}
        return self._result(info), info

    def step(self, action):
        a = torch.tensor(np.asarray(action, dtype=np.float64).reshape(-1)[: self._b.action_dim], dtype=torch.float64)
        self._b.step(a)
        it = self._b.info[0].cpu().numpy()
        boxes = self._boxes()
        self.box_clearance_statuses = self._statuses(boxes)
        info = {"state": (round(float(it[0]), 2), round(float(it[1]), 2), round(float(it[2]), 2)), "total_work": float(it[3]),
                "collision reward": float(it[4]), "diff_reward": float(it[5]), "box_completed_reward": float(it[6]), "obs": boxes,
                "box_completed_statuses": self.box_clearance_statuses, "box_count": int(it[7]), "ministeps": float(it[8]),
                "low_level_observation": self._low_dim(boxes)}
        return (self._result(info), float(self._b.reward[0].item()), bool(self._b.terminated[0].item()),
                bool(self._b.truncated[0].item()), info)

    def render(self, mode="human", close=False):
        raise NotImplementedError("rendering (pygame) is outside the accelerated path")

    def close(self):
        self._b.close()
