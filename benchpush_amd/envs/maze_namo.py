"""maze-NAMO-v0 on MI355X: batched tensor environment + the reference-shaped single-env adapter.

Reference: benchpush/environments/maze_NAMO/maze_NAMO_env.py (MazeNAMO).  Same engine as ship-ice: the robot is a
kinematic body of five shapes (outline + four wheels, robot.py:77-118), boxes are dynamic squares, walls are static
Segment(radius 0.5) shapes; 400 sub-steps per step, reward from the work term and the BFS goal map, observation
uint8 [4, 192, 192] = rotated ego views of [footprint, boxes, walls, goal map].

The reference draws a new random box layout from the unseeded global ``random`` at every reset; here layout t comes
from ``random.Random(base_seed + t)`` and env e plays layout (global_env_id + episode) % T.
"""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib
from ..config import default_cfg, maze_physics_params, maze_walls, merge_user_cfg
from ..gym_shim import Env, spaces
from ..maze_scenario import generate_layout
from ..scenario import poly_centroid
from .ship_ice import BatchedShipIceEnv, _ptr

__all__ = ["BatchedMazeEnv", "MazeNAMO", "default_layouts"]

MAZE_INFO_KEYS = ["x", "y", "theta", "total_work", "work", "collision_reward", "scaled_collision_reward", "dist_increment_reward",
                  "trial_success", "boundary_violated", "wall_collision", "total_ke", "total_impulse", "n_post_solve",
                  "n_contact_pts", "n_first_contact"]


def _maze_cfg(cfg):
    c = merge_user_cfg(default_cfg("maze_namo"), cfg)
    if c.maze_version == 1:      # maze_NAMO_env.py:68-73
        c.env = c.env1
    elif c.maze_version == 2:
        c.env = c.env2
    else:
        raise Exception("Invalid Maze Version!")
    return c


def default_layouts(cfg, num_layouts, base_seed=0):
    c = _maze_cfg(cfg) if "env" not in cfg else cfg
    walls = maze_walls(c)
    return [generate_layout(c, walls, base_seed + t) for t in range(num_layouts)]


class BatchedMazeEnv(BatchedShipIceEnv):
    """E independent maze-NAMO environments on one GPU (reset / step / world_polys / ... as BatchedShipIceEnv)."""

    def __init__(self, num_envs, cfg=None, layouts=None, device="cuda:0", env_id_offset=0, num_layouts=64, base_seed=0):
        if not torch.cuda.is_available():
            raise _lib.BpError("BatchedMazeEnv needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.L = _lib.load()
        self.cfg = _maze_cfg(cfg)
        self.num_envs = int(num_envs)
        self.env_id_offset = int(env_id_offset)
        self.device = torch.device(device)
        self.params = maze_physics_params(self.cfg)
        self.goal = (self.cfg.env.goal_x, self.cfg.env.goal_y)
        self.max_yaw_rate_step = (math.pi / 2) / 15
        self.walls = maze_walls(self.cfg)
        if layouts is None:
            layouts = [generate_layout(self.cfg, self.walls, base_seed + t) for t in range(num_layouts)]
        self.layouts = layouts
        self.trials = layouts
        rv = self.cfg.robot.vertices
        head = ((rv[0][0] + rv[3][0]) / 2, (rv[0][1] + rv[3][1]) / 2)   # maze_NAMO_env.py:94-95
        tail = ((rv[1][0] + rv[2][0]) / 2, (rv[1][1] + rv[2][1]) / 2)
        bcfg = _lib.make_config(self.params, rv, head, tail, env_kind=_lib.ENV_MAZE, wheel_vertices=self.cfg.robot.wheel_vertices,
                                obstacle_size=self.cfg.obstacle_size)
        self.h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self.L, None, self.L.bp_create(C.byref(bcfg), self.num_envs, int(env_id_offset), dev_index, C.byref(self.h)), "bp_create")
        nbox = len(layouts[0]["centres"])
        if any(len(l["centres"]) != nbox for l in layouts):
            raise ValueError("all layouts must hold the same number of boxes")
        centres = np.ascontiguousarray(np.stack([np.asarray(l["centres"], np.float64).reshape(nbox, 2) for l in layouts]))
        walls = np.ascontiguousarray(self.walls, np.float64)
        # one start pose per layout: the fixed pose, or the layout's draw of cfg.random_start (the reference re-draws per episode from
        # python's unseeded global generator; here episode k plays layout (env + k) % T, whose start was drawn from Random(base_seed + t))
        start = np.ascontiguousarray(np.stack([np.asarray(l["start"], np.float64).reshape(3) for l in layouts]))
        _lib.check(self.L, self.h, self.L.bp_load_maze(self.h, len(layouts), nbox, centres.ctypes.data_as(C.c_void_p), len(walls),
                                                       walls.ctypes.data_as(C.c_void_p), start.ctypes.data_as(C.c_void_p)), "bp_load_maze")
        self._alloc_io()

    def goal_map(self):
        """info['goal_dt'] of the reference: un-normalised wavefront distances, numpy [grid_h, grid_w]."""
        gh, gw = C.c_int32(), C.c_int32()
        _lib.check(self.L, self.h, self.L.bp_get_goal_map(self.h, None, C.byref(gh), C.byref(gw)), "bp_get_goal_map")
        out = np.zeros((gh.value, gw.value), np.float64)
        _lib.check(self.L, self.h, self.L.bp_get_goal_map(self.h, out.ctypes.data_as(C.c_void_p), None, None), "bp_get_goal_map")
        return out


class MazeNAMO(Env):
    """Reference-shaped single environment (E = 1): reset()/step() returns and info keys of maze_NAMO_env.py:325-485."""

    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 4}

    def __init__(self, cfg=None, layouts=None, device="cuda:0", num_layouts=64, base_seed=0):
        super().__init__()
        self._b = BatchedMazeEnv(1, cfg=cfg, layouts=layouts, device=device, num_layouts=num_layouts, base_seed=base_seed)
        self.cfg = self._b.cfg
        self.beta = 1.5
        self.k = 2
        self.k_increment = 150
        self.episode_idx = None
        self.path = None
        self.low_dim_state = self.cfg.low_dim_state
        self.env_max_trial = 4000
        self.max_linear_speed = 1.0
        self.min_linear_speed = 0.0
        self.max_yaw_rate_step = (np.pi / 2) / 15
        self.action_space = spaces.Box(low=-1, high=1, dtype=np.float64)
        self.observation_shape = self._b.obs_shape
        if self.low_dim_state:  # maze_NAMO_env.py:106-112
            self.fixed_trial_idx = self.cfg.fixed_trial_idx
            n = (self.cfg.num_obstacles + 1) * 2 if self.cfg.randomize_obstacles else 8
            self.observation_space = spaces.Box(low=-10, high=30, shape=(n,), dtype=np.float64)
        else:
            self.observation_space = spaces.Box(low=0, high=255, shape=self.observation_shape, dtype=np.uint8)
        self.goal = self._b.goal
        self.total_work = [0, []]
        self.wall_collision = False
        self.t = 0
        self._goal_dt = self._b.goal_map()

    def _polys(self):
        verts, cnt = self._b.world_polys()
        verts, cnt = verts[0].cpu().numpy(), cnt[0].cpu().numpy()
        n0 = 1 + len(self.cfg.robot.wheel_vertices)
        nbox = len(self._b.layouts[0]["centres"])
        return [verts[i, : cnt[i]].copy() for i in range(n0, n0 + nbox)]

    def _low_dim(self, obstacles, it):
        """generate_observation_low_dim (maze_NAMO_env.py:488-504): <robot x, y, |centroid| of obstacles 1..n-1 at pairs 1..n-1>; like
        the reference's loop, obstacle 0 is never written and the last pair stays zero."""
        out = np.zeros((len(obstacles) + 1) * 2)
        out[0], out[1] = float(it[0]), float(it[1])
        for i in range(1, len(obstacles)):
            out[2 * i: 2 * i + 2] = poly_centroid(obstacles[i])
        return out

    def _observation(self, obstacles, it):
        return self._low_dim(obstacles, it) if self.low_dim_state else self._b.obs[0].cpu().numpy()

    def reset(self, seed=None, options=None):
        self.episode_idx = 0 if self.episode_idx is None else self.episode_idx + 1
        if self.episode_idx:
            self._b.check_errors()   # capacity flags of the episode that just ended (raises BpError)
        self._b.reset()
        self.t = 0
        self.total_work = [0, []]
        self.wall_collision = False
        it = self._b.info[0].cpu().numpy()
        obstacles = self._polys()
        self.obstacles = obstacles
        info = {"state": (round(float(it[0]), 2), round(float(it[1]), 2), round(float(it[2]), 2)), "total_work": self.total_work[0],
                "obs": obstacles, "box_count": 0, "goal_dt": self._goal_dt, "m_to_pix_scale": self.cfg.occ.m_to_pix_scale}
        return self._observation(obstacles, it), info

    def step(self, action):
        self.t += 1
        a = torch.tensor([float(np.asarray(action, dtype=np.float64).reshape(-1)[0])], dtype=torch.float64)
        self._b.step(a)
        it = self._b.info[0].cpu().numpy()
        reward = float(self._b.reward[0].item())
        terminated = bool(self._b.terminated[0].item())
        obstacles = self._polys()
        self.obstacles = obstacles
        self.total_work[0] = float(it[3])
        self.total_work[1].append(float(it[4]))
        self.wall_collision = bool(it[10])
        info = {"state": (round(float(it[0]), 2), round(float(it[1]), 2), round(float(it[2]), 2)), "total_work": self.total_work[0],
                "collision reward": float(it[5]), "scaled collision reward": float(it[6]), "dist increment reward": float(it[7]),
                "trial_success": bool(it[8]), "obs": obstacles}
        if self.cfg.log_obs:
            self.log_observation()
        return self._observation(obstacles, it), reward, terminated, False, info

    def log_observation(self):
        """`cfg.log_obs` (maze_NAMO_env.py:482-483, 539-595): <cfg.output_dir>/t<episode_idx>/<t>_footprint, _movable_obs, _fixed_obs, _local_distance_map (the four
        observation channels) and _distance_map (the global goal-distance map: wavefront distances over their maximum, walls and unreached cells 1.0)."""
        from ..obs_log import dump_channels
        o = self._b.obs[0].cpu().numpy()               # [footprint, boxes, walls, distance] (occupancy_map.py:142-202)
        g = np.asarray(self._goal_dt, np.float64)
        gn = g / g.max() if g.max() > 0 else g.copy()
        gn[g == 0] = 1.0
        return dump_channels(self.cfg.output_dir, self.episode_idx, self.t,
                             {"footprint": o[0], "movable_obs": o[1], "fixed_obs": o[2], "distance_map": gn, "local_distance_map": o[3]})

    def update_path(self, new_path, scatter=False):
        self.path = new_path

    def render(self, mode="human", close=False):
        raise NotImplementedError("rendering (pygame) is outside the accelerated path")

    def close(self):
        self._b.close()
