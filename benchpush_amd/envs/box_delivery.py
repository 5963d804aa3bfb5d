"""box-delivery-v0 on MI355X: batched tensor environment + the reference-shaped single-env adapter.

Reference: benchpush/environments/box_delivery/box_delivery_env.py (BoxDeliveryEnv) with ``agent.action_type: 'heading'`` (what
its PPO/SAC baselines use, config.yaml:170): one step = PositionController waypoints + execute_robot_path (a variable number
of 2 ms sim steps under the DP controller) + step_simulation_until_still, rewards from spfa path-length deltas of every box,
boxes inside the receptacle are removed, observation uint8 [224, 224, 4] (channels last).

Episodes: the reference keeps one ``np.random.RandomState(cfg.misc.random_seed)`` per env and draws start pose / boxes at every
reset; here the same stream generates ``num_trials`` consecutive episodes once and env e plays episode
``(global_env_id + episode) % num_trials``.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..box_delivery_scenario import box_delivery_params, box_delivery_physics_params, generate_trials
from ..config import default_cfg, merge_user_cfg
from ..gym_shim import Env, spaces
from ..scenario import poly_centroid
from .ship_ice import BatchedShipIceEnv, _ptr

__all__ = ["BatchedBoxDeliveryEnv", "BoxDeliveryEnv", "BD_INFO_KEYS"]
BD_INFO_KEYS = _lib.BD_INFO_KEYS


def _bd_cfg(cfg):
    c = merge_user_cfg(default_cfg("box_delivery"), cfg)
    if c.agent.action_type not in ("heading", "position", "velocity"):
        raise ValueError("agent.action_type must be heading, position or velocity")
    if c.teleop_mode:
        raise NotImplementedError("teleop mode (keyboard / pygame) is outside the accelerated path")
    return c


class BatchedBoxDeliveryEnv(BatchedShipIceEnv):
    """E independent box-delivery environments on one GPU: reset(mask) / step(actions) with device tensors."""

    def __init__(self, num_envs, cfg=None, trials=None, device="cuda:0", env_id_offset=0, num_trials=32, seed=None, bd_overrides=None):
        if not torch.cuda.is_available():
            raise _lib.BpError("BatchedBoxDeliveryEnv needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.L = _lib.load()
        self.cfg = _bd_cfg(cfg)
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self.params = box_delivery_physics_params(self.cfg)
        self.bd_params = box_delivery_params(self.cfg)
        if bd_overrides:   # constants of the reference that are not in its config (e.g. STEP_LIMIT, box_delivery_env.py:62): tests shorten them
            self.bd_params.update(bd_overrides)
        if trials is None:
            trials = generate_trials(self.cfg, num_trials, seed)
        self.trials = trials
        nbox = len(trials[0]["boxes"])
        ns = max(len(t["statics"][1]) for t in trials)
        if any(len(t["boxes"]) != nbox for t in trials):
            raise ValueError("all trials must hold the same number of boxes")
        self.bd_params["num_boxes"] = nbox
        self.nbox = nbox
        bcfg = _lib.make_bd_config(self.params, self.bd_params, self.cfg)
        self.h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self.L, None, self.L.bp_bd_create(C.byref(bcfg), self.num_envs, int(env_id_offset), dev_index, C.byref(self.h)), "bp_bd_create")
        c = lambda a, dt: np.ascontiguousarray(a, dt)
        starts = c(np.stack([t["start"] for t in trials]), np.float64)
        boxes = c(np.stack([t["boxes"] for t in trials]), np.float64)
        def pad(a, fill=0):   # trials may hold different numbers of columns: unused slots have vertex count 0
            out = np.full((ns,) + a.shape[1:], fill, a.dtype)
            out[: len(a)] = a
            return out
        sv = c(np.stack([pad(t["statics"][0]) for t in trials]), np.float64)
        sc = c(np.stack([pad(t["statics"][1]) for t in trials]), np.int32)
        sp = c(np.stack([pad(t["statics"][2]) for t in trials]), np.float64)
        sr = c(np.stack([pad(t["statics"][3]) for t in trials]), np.float64)
        st = c(np.stack([pad(t["statics"][4], 3) for t in trials]), np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self.L, self.h, self.L.bp_bd_load(self.h, len(trials), nbox, p(starts), p(boxes), ns, p(sv), p(sc), p(sp), p(sr), p(st)), "bp_bd_load")
        self._alloc_io()
        lp = self.L.bp_obs_height(self.h)
        self.obs_shape = (lp, lp, 4)
        self.obs = torch.zeros((self.num_envs,) + self.obs_shape, dtype=torch.uint8, device=self.device)
        self.action_dim = 2 if self.cfg.agent.action_type == "velocity" else 1
        self._actions = torch.zeros(self.num_envs * self.action_dim, dtype=torch.float64, device=self.device)

    def step(self, actions):
        """actions: [E] heading in [-1, 1] / position index, or [E, 2] = (linear, angular) speed for 'velocity'."""
        self._actions.copy_(actions.reshape(-1).to(self.device), non_blocking=True)
        _lib.check(self.L, self.h, self.L.bp_step(self.h, _ptr(self._actions), _ptr(self.obs), _ptr(self.reward), _ptr(self.terminated),
                                                 _ptr(self.truncated), _ptr(self.info), self._stream()), "bp_step")
        return self.obs, self.reward, self.terminated, self.truncated, self.info

    def stragglers(self):
        """(env steps finished by the second pass of the two-pass step, env steps whose path / until-still loop ran into STEP_LIMIT), cumulative since load."""
        out = np.zeros(2, np.uint32)
        _lib.check(self.L, self.h, self.L.bp_bd_get_stragglers(self.h, out.ctypes.data_as(C.c_void_p)), "bp_bd_get_stragglers")
        return int(out[0]), int(out[1])

    def cycle_skips(self):
        """(exact recurrences found in execute_robot_path, sim steps they skipped), cumulative since load (bp_bd_get_cycle_skips)."""
        out = np.zeros(2, np.uint32)
        _lib.check(self.L, self.h, self.L.bp_bd_get_cycle_skips(self.h, out.ctypes.data_as(C.c_void_p)), "bp_bd_get_cycle_skips")
        return int(out[0]), int(out[1])

    def maps(self, trial=0):
        dims = np.zeros(6, np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self.L, self.h, self.L.bp_bd_get_maps(self.h, trial, p(dims), None, None, None, None, None), "bp_bd_get_maps")
        SH, SW = int(dims[2]), int(dims[3])
        out = dict(dims=dims, cspace=np.zeros((SH, SW), np.uint8), cspace_thin=np.zeros((SH, SW), np.uint8), edt=np.zeros((SH, SW, 2), np.uint16),
                   recept=np.zeros((SH, SW), np.float32), small_free=np.zeros((SH, SW), np.uint8))
        _lib.check(self.L, self.h, self.L.bp_bd_get_maps(self.h, trial, p(dims), p(out["cspace"]), p(out["cspace_thin"]), p(out["edt"]), p(out["recept"]),
                                                        p(out["small_free"])), "bp_bd_get_maps")
        return out

    def box_state(self):
        """(alive uint8 [E, 24], waypoints [E, 64, 3], number of waypoints [E]) of the last step (host copies)."""
        alive = np.zeros((self.num_envs, _lib.BD_MAXBOX), np.uint8)
        wp = np.zeros((self.num_envs, _lib.BD_MAXWP, 3), np.float64)
        nwp = np.zeros(self.num_envs, np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self.L, self.h, self.L.bp_bd_get_state(self.h, p(alive), p(wp), p(nwp)), "bp_bd_get_state")
        return alive, wp, nwp


def low_dim_observation(polys):
    """generate_observation_low_dim (box_delivery_env.py:1025-1037, area_clearing.py:908-919): |centroid| of each polygon, flattened."""
    out = np.zeros(len(polys) * 2)
    for i, p in enumerate(polys):
        out[2 * i: 2 * i + 2] = poly_centroid(p)
    return out


class BoxDeliveryEnv(Env):
    """Reference-shaped single environment (E = 1): reset()/step() returns and info keys of box_delivery_env.py:578-830."""

    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 4}

    def __init__(self, cfg=None, trials=None, device="cuda:0", num_trials=32):
        super().__init__()
        self._b = BatchedBoxDeliveryEnv(1, cfg=cfg, trials=trials, device=device, num_trials=num_trials)
        self.cfg = self._b.cfg
        from ..obs_log import refuse_render_log_obs
        refuse_render_log_obs(self.cfg, "box-delivery-v0")
        self.num_boxes = self._b.nbox
        lp = self._b.obs_shape[0]
        if self.cfg.agent.action_type == "velocity":     # box_delivery_env.py:157-162
            self.action_space = spaces.Box(low=-1, high=1, shape=(2,), dtype=np.float32)
        elif self.cfg.agent.action_type == "heading":
            self.action_space = spaces.Box(low=-1, high=1, shape=(1,), dtype=np.float32)
        else:
            self.action_space = spaces.Box(low=0, high=lp * lp, dtype=np.float32)
        self.observation_shape = self._b.obs_shape
        self.low_dim_state = self.cfg.low_dim_state
        if self.low_dim_state:                            # box_delivery_env.py:166-169
            self.fixed_trial_idx = self.cfg.fixed_trial_idx
            self.observation_space = spaces.Box(low=-10, high=30, shape=(self.num_boxes * 2,), dtype=np.float32)
        else:
            self.observation_space = spaces.Box(low=0, high=255, shape=self.observation_shape, dtype=np.uint8)
        self.episode_idx = None
        self.t = 0
        self.box_clearance_statuses = [False] * self.num_boxes
        self.receptacle_position = (self._b.bd_params["recept_x"], self._b.bd_params["recept_y"])
        self.goal_points = [self.receptacle_position]

    def _boxes(self):
        verts, cnt = self._b.world_polys()
        alive, _, _ = self._b.box_state()
        verts, cnt = verts[0].cpu().numpy(), cnt[0].cpu().numpy()
        return [verts[6 + k, : cnt[6 + k]].copy() for k in range(self.num_boxes) if alive[0, k]], alive[0, : self.num_boxes]

    def _info(self, it, boxes, alive):
        self.box_clearance_statuses = [not bool(a) for a in alive]
        return {"state": (round(float(it[0]), 2), round(float(it[1]), 2), round(float(it[2]), 2)), "cumulative_distance": float(it[3]),
                "cumulative_boxes": int(it[4]), "cumulative_reward": float(it[5]), "total_work": float(it[6]), "obs": boxes,
                "box_completed_statuses": self.box_clearance_statuses, "goal_positions": self.goal_points, "ministeps": float(it[7]),
                "inactivity": int(it[8])}

    def reset(self, seed=None, options=None):
        self.episode_idx = 0 if self.episode_idx is None else self.episode_idx + 1
        if self.episode_idx:
            self._b.check_errors()   # capacity flags of the episode that just ended (raises BpError)
        self._b.reset()
        self.t = 0
        boxes, alive = self._boxes()
        # low_dim_state: reset() returns <|centroid| of every box> (box_delivery_env.py:606-607,1025-1037); step() always returns the
        # image observation (box_delivery_env.py:807)
        obs = low_dim_observation(boxes) if self.low_dim_state else self._b.obs[0].cpu().numpy()
        return obs, self._info(self._b.info[0].cpu().numpy(), boxes, alive)

    def step(self, action):
        self.t += 1
        a = torch.tensor(np.asarray(action, dtype=np.float64).reshape(-1)[: self._b.action_dim], dtype=torch.float64)
        self._b.step(a)
        boxes, alive = self._boxes()
        info = self._info(self._b.info[0].cpu().numpy(), boxes, alive)
        return (self._b.obs[0].cpu().numpy(), float(self._b.reward[0].item()), bool(self._b.terminated[0].item()),
                bool(self._b.truncated[0].item()), info)

    def render(self, mode="human", close=False):
        raise NotImplementedError("rendering (pygame) is outside the accelerated path")

    def close(self):
        self._b.close()
