"""ship-ice-v0 on MI355X: batched tensor environment + the reference-shaped single-env adapter.

Reference: benchpush/environments/ship_ice_nav/ship_ice_env.py (ShipIceEnv).  ``BatchedShipIceEnv`` is the native
surface (device tensors in / out, all envs stepped by one kernel launch pair); ``ShipIceEnv`` mirrors the reference
class for drop-in use by BasePolicy / BaseMetric code (same constructor, reset()/step() returns, info keys,
attributes ``cfg``, ``goal``, ``max_yaw_rate_step``, ``action_space``, ``observation_space``, ``unwrapped``).

PyTorch is used only for device memory and streams; all compute is in libbenchpush_hip.so (C ABI).
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from .. import _lib
from ..config import DotDict, default_cfg, merge_user_cfg, ship_ice_physics_params
from ..gym_shim import Env, spaces
from ..scenario import generate_ice_field, load_experiment, pack_trials

__all__ = ["BatchedShipIceEnv", "ShipIceEnv", "default_trials"]

_CONC_GEN = {  # synthetic generator radii per concentration (floe count ~ BASELINE.json configs)
    0.1: dict(min_r=0.40, max_r=0.58), 0.2: dict(min_r=0.40, max_r=0.58), 0.3: dict(min_r=0.40, max_r=0.58),
    0.4: dict(min_r=0.40, max_r=0.58), 0.5: dict(min_r=0.40, max_r=0.58),
}


def default_trials(concentration, num_trials, base_seed=0, goal_y=9.0):
    """Synthetic stand-in for experiments_<conc>_100_r06_d40x12.pk (missing blobs): trial i <- seed base_seed+i."""
    kw = _CONC_GEN.get(round(float(concentration), 2), dict(min_r=0.40, max_r=0.58))
    return [generate_ice_field(float(concentration), base_seed + i, goal_y=goal_y, **kw) for i in range(num_trials)]


def experiment_file(concentration, directory):
    """The reference's file name for a concentration (ship_ice_env.py:76): ice_environments/experiments_<c*100>_100_r06_d40x12.pk."""
    return os.path.join(directory, "experiments_" + str(int(concentration * 100)) + "_100_r06_d40x12.pk")


def resolve_trials(cfg, num_trials=100, base_seed=0):
    """Trials of an environment, like the reference's constructor (ship_ice_env.py:74-80): the pickled experiment file of the configured
    concentration when one is available -- looked up in ``cfg.ice_environments_dir``, ``$BENCHPUSH_ICE_DIR`` and
    ``benchpush_amd/ice_environments`` -- else the synthetic stand-in (the reference checkout ships the files as missing LFS blobs)."""
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ice_environments")
    for d in (cfg.get("ice_environments_dir", None) if hasattr(cfg, "get") else None, os.environ.get("BENCHPUSH_ICE_DIR"), here):
        if d and os.path.isfile(experiment_file(cfg.concentration, d)):
            exp = load_experiment(experiment_file(cfg.concentration, d), cfg.concentration)
            return [exp[k] for k in sorted(exp)]
    return default_trials(cfg.concentration, num_trials, base_seed, goal_y=cfg.goal_y)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _BatchedBase:
    """Tensor-in / tensor-out wrapper of one bp_handle (shared by the ship-ice and maze environments)."""

    def _alloc_io(self):
        self.nb_cap = self.L.bp_nb_cap(self.h)
        self.obs_shape = (4, self.L.bp_obs_height(self.h), self.L.bp_obs_width(self.h))
        E, dv = self.num_envs, self.device
        self.obs = torch.zeros((E,) + self.obs_shape, dtype=torch.uint8, device=dv)
        self.reward = torch.zeros(E, dtype=torch.float64, device=dv)
        self.terminated = torch.zeros(E, dtype=torch.uint8, device=dv)
        self.truncated = torch.zeros(E, dtype=torch.uint8, device=dv)
        self.info = torch.zeros((E, _lib.INFO_COUNT), dtype=torch.float64, device=dv)
        self._actions = torch.zeros(E, dtype=torch.float64, device=dv)


class BatchedShipIceEnv(_BatchedBase):
    """E independent ship-ice environments on one GPU.

    reset(mask) / step(actions) follow ShipIceEnv.reset / .step (ship_ice_env.py:223-355) for every env at once.
    Trial selection generalises ``episode_idx % len(experiment)`` (:188) to ``(global_env_id + episode_idx) % T``.
    """

    def __init__(self, num_envs, cfg=None, trials=None, device="cuda:0", env_id_offset=0, num_trials=100, base_seed=0):
        if not torch.cuda.is_available():
            raise _lib.BpError("BatchedShipIceEnv needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.L = _lib.load()
        self.cfg = merge_user_cfg(default_cfg("ship_ice"), cfg)
        assert self.cfg.concentration in [0.1, 0.2, 0.3, 0.4, 0.5]  # ship_ice_env.py:75
        self.num_envs = int(num_envs)
        self.env_id_offset = int(env_id_offset)
        self.device = torch.device(device)
        self.params = ship_ice_physics_params(self.cfg)
        self.goal = (0, self.cfg.goal_y)
        self.max_yaw_rate_step = (math.pi / 2) / 7
        if trials is None:
            trials = resolve_trials(self.cfg, num_trials, base_seed)
        if self.cfg.low_dim_state:  # ship_ice_env.py:190-191 pins one trial
            trials = [trials[self.cfg.fixed_trial_idx]]
        self.trials = trials
        bcfg = _lib.make_config(self.params, self.cfg.ship.vertices, self.cfg.ship.head, self.cfg.ship.tail)
        self.h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self.L, None, self.L.bp_create(C.byref(bcfg), self.num_envs, int(env_id_offset), dev_index, C.byref(self.h)),
                   "bp_create")
        pk = pack_trials(trials, max_verts=_lib.MAXV)
        self._pk = pk
        T, F, V = pk["verts"].shape[:3]
        _lib.check(self.L, self.h, self.L.bp_load_scenarios(
            self.h, T, F, V, pk["verts"].ctypes.data_as(C.c_void_p), pk["counts"].ctypes.data_as(C.c_void_p),
            pk["centres"].ctypes.data_as(C.c_void_p), pk["starts"].ctypes.data_as(C.c_void_p),
            pk["nfloes"].ctypes.data_as(C.c_void_p)), "bp_load_scenarios")
        self._alloc_io()

    # -- helpers -----------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "h", None):
            self.L.bp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- API -----------------------------------------------------------------------------------------------
    def reset(self, mask=None):
        """Reset the envs selected by ``mask`` (uint8/bool tensor [E]; None = all). Returns (obs, info) views."""
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        _lib.check(self.L, self.h, self.L.bp_reset(self.h, _ptr(m), _ptr(self.obs), _ptr(self.info), self._stream()), "bp_reset")
        return self.obs, self.info

    def step(self, actions):
        """actions: tensor [E] in [-1, 1] (any float dtype).  Returns (obs, reward, terminated, truncated, info)."""
        self._actions.copy_(actions.reshape(-1).to(self.device), non_blocking=True)
        _lib.check(self.L, self.h, self.L.bp_step(self.h, _ptr(self._actions), _ptr(self.obs), _ptr(self.reward),
                                                 _ptr(self.terminated), _ptr(self.truncated), _ptr(self.info),
                                                 self._stream()), "bp_step")
        return self.obs, self.reward, self.terminated, self.truncated, self.info

    def step_physics(self, actions):
        self._actions.copy_(actions.reshape(-1).to(self.device), non_blocking=True)
        _lib.check(self.L, self.h, self.L.bp_step_physics(self.h, _ptr(self._actions), _ptr(self.reward), _ptr(self.terminated),
                                                         _ptr(self.truncated), _ptr(self.info), self._stream()), "bp_step_physics")
        return self.reward, self.terminated, self.truncated, self.info

    def observe(self, mask=None):
        m = None if mask is None else mask.to(device=self.device, dtype=torch.uint8).contiguous()
        _lib.check(self.L, self.h, self.L.bp_observe(self.h, _ptr(m), _ptr(self.obs), self._stream()), "bp_observe")
        return self.obs

    def observe_global(self, mask=None):
        """Planner observation (cfg.egocentric_obs: false): uint8 [E, 2, map_h/0.2, map_w/0.2]."""
        gh, gw = int(self.cfg.occ.map_height / 0.2), int(self.cfg.occ.map_width / 0.2)
        if getattr(self, "_gobs", None) is None:
            self._gobs = torch.zeros((self.num_envs, 2, gh, gw), dtype=torch.uint8, device=self.device)
        m = None if mask is None else mask.to(device=self.device, dtype=torch.uint8).contiguous()
        _lib.check(self.L, self.h, self.L.bp_observe_global(self.h, _ptr(m), _ptr(self._gobs), self._stream()), "bp_observe_global")
        return self._gobs

    def world_polys(self):
        """info['obs'] for every env: (verts [E, nb_cap, 20, 2] f64, counts [E, nb_cap] i32); index 0 is the ship."""
        out = torch.zeros((self.num_envs, self.nb_cap, _lib.MAXV, 2), dtype=torch.float64, device=self.device)
        cnt = torch.zeros((self.num_envs, self.nb_cap), dtype=torch.int32, device=self.device)
        _lib.check(self.L, self.h, self.L.bp_get_world_polys(self.h, _ptr(out), _ptr(cnt), self._stream()), "bp_get_world_polys")
        return out, cnt

    def body_state(self):
        out = torch.zeros((self.num_envs, self.nb_cap, 9), dtype=torch.float64, device=self.device)
        _lib.check(self.L, self.h, self.L.bp_get_body_state(self.h, _ptr(out), self._stream()), "bp_get_body_state")
        return out

    def low_dim_obs(self):
        out = torch.zeros((self.num_envs, self.nb_cap - 1, 2), dtype=torch.float64, device=self.device)
        _lib.check(self.L, self.h, self.L.bp_get_low_dim_obs(self.h, _ptr(out), self._stream()), "bp_get_low_dim_obs")
        return out

    def cost_maps(self, scale, m, n, alpha=10.0, ship_mass=1.0, horizon=None, margin=1, ship_pos_y=None, vs=1.0, out=None):
        """Planner cost maps of every env (CostMap(...).update(info['obs'], ship_pos_y, vs).cost_map, common/cost_map.py:27-126):
        float64 [E, int(m*scale), int(n*scale)] on the device.  ship_pos_y: [E] tensor in cost-map units, or None for 0."""
        H, W = int(m * scale), int(n * scale)
        if out is None:
            out = torch.empty((self.num_envs, H, W), dtype=torch.float64, device=self.device)
        cfg = _lib.BpCostmapConfig(scale=float(scale), m=int(m), n=int(n), alpha=float(alpha), ship_mass=float(ship_mass),
                                   horizon=float(horizon or 0.0), margin=int(margin), pad_=0)
        spy = None
        if ship_pos_y is not None:
            spy = torch.as_tensor(ship_pos_y, dtype=torch.float64).to(self.device).reshape(self.num_envs).contiguous()
        _lib.check(self.L, self.h, self.L.bp_costmap_update(self.h, C.byref(cfg), _ptr(spy) if spy is not None else None, float(vs),
                                                            _ptr(out), self._stream()), "bp_costmap_update")
        return out

    def episode_metrics(self):
        """On-device ShipIceMetric: (rows [E, 6] float64 = efficiency, effort, episode reward, success, episode length, total_work of
        each env's most recently finished episode; counts [E] int32 = episodes finished so far).  Device tensors; rows of envs with
        count 0 are zero.  This is the block benchpush_amd.parallel.allgather_episode_metrics carries between GPUs."""
        rows = torch.zeros((self.num_envs, 6), dtype=torch.float64, device=self.device)
        cnt = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        _lib.check(self.L, self.h, self.L.bp_get_episode_metrics(self.h, _ptr(rows), _ptr(cnt), self._stream()), "bp_get_episode_metrics")
        return rows, cnt

    def episode_history(self):
        """The episode lists of BaseMetric (base_metric.py:12-16) kept on the device: (ring [E, 8, 6] float64: the last 8 finished episodes of each
        env, episode n in slot n % 8; sums [E, 6] float64: the six row fields summed over all finished episodes; counts [E] int32).  No host
        round trip per step is needed to follow the episodes: see episode_lists() and benchpush_amd.parallel.gather_episode_sums."""
        ring = torch.zeros((self.num_envs, _lib.EPM_RING, _lib.EPM_COUNT), dtype=torch.float64, device=self.device)
        sums = torch.zeros((self.num_envs, _lib.EPM_COUNT), dtype=torch.float64, device=self.device)
        cnt = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        _lib.check(self.L, self.h, self.L.bp_get_episode_history(self.h, _ptr(ring), _ptr(sums), _ptr(cnt), self._stream()), "bp_get_episode_history")
        return ring, sums, cnt

    def episode_lists(self):
        """Per env, the rows of its finished episodes in order (the last 8 at most): a list of E float64 arrays [n_e, 6] (host)."""
        ring, _, cnt = self.episode_history()
        ring, cnt = ring.cpu().numpy(), cnt.cpu().numpy()
        out = []
        for e in range(self.num_envs):
            n = int(cnt[e])
            out.append(np.stack([ring[e, k % _lib.EPM_RING] for k in range(max(0, n - _lib.EPM_RING), n)]) if n else np.zeros((0, _lib.EPM_COUNT)))
        return out

    def start_uniform(self, env, episode):
        """The uniform of the counter RNG that places env's ship in `episode` when cfg.random_start is set (bp_start_uniform)."""
        return float(self.L.bp_start_uniform(int(self.params["start_seed"]), int(self.env_id_offset) + int(env), int(episode)))

    def num_bodies(self):
        out = np.zeros(self.num_envs, np.int32)
        _lib.check(self.L, self.h, self.L.bp_get_num_bodies(self.h, out.ctypes.data_as(C.c_void_p)), "bp_get_num_bodies")
        return out

    def check_errors(self):
        out = np.zeros(self.num_envs, np.int32)
        rc = self.L.bp_check_errors(self.h, out.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise _lib.BpError("capacity overflow in envs %s: %s" % (np.nonzero(out)[0][:8].tolist(),
                                                                     self.L.bp_last_error(self.h).decode()))

    def set_resettle(self, on=True):
        """Re-run the 1000 settle sub-steps on every reset instead of copying the settled per-trial template."""
        self.L.bp_set_resettle(self.h, int(on))

    def step_cycles(self):
        """Shader cycles each env's wavefront spent in the last step (numpy uint64 [E])."""
        out = np.zeros(self.num_envs, np.uint32)
        _lib.check(self.L, self.h, self.L.bp_get_step_cycles(self.h, out.ctypes.data_as(C.c_void_p)), "bp_get_step_cycles")
        return out.astype(np.uint64) << 8

    def clock_stamps(self):
        """uint64 [8, 2]: per XCD the latest (shader-clock counter, 100 MHz reference counter) pair stamped after a physics launch by a thread of that
        XCD (zeros: none yet).  `clock_hz_between` turns two readings into the clock the chip held in between."""
        out = np.zeros((8, 2), np.uint64)
        _lib.check(self.L, self.h, self.L.bp_get_clock_stamps(self.h, out.ctypes.data_as(C.c_void_p)), "bp_get_clock_stamps")
        return out

    @staticmethod
    def clock_hz_between(c0, c1, with_span=False):
        """Shader clock between two `clock_stamps()` readings: both counters of ONE XCD (the counters of different XCDs are not synchronised).  Only the
        thread of env 0 stamps, so most XCDs are refreshed rarely: an XCD whose first stamp is old (load, warm-up) would bring idle clocks into the span.  The
        XCD chosen is therefore the one whose FIRST stamp is the newest -- closest to the start of the timed region -- among those stamped again afterwards.
        Returns (hz, xcd) or (None, None) if no XCD was stamped before both readings; with_span=True appends the reference-time span in seconds."""
        best = None
        for x in range(8):
            if c0[x, 1] == 0 or c1[x, 1] <= c0[x, 1]:
                continue
            if best is None or int(c0[x, 1]) > int(c0[best, 1]):
                best = x
        if best is None:
            return (None, None, None) if with_span else (None, None)
        span = int(c1[best, 1]) - int(c0[best, 1])
        hz = (int(c1[best, 0]) - int(c0[best, 0])) / span * 1e8
        return (hz, best, span / 1e8) if with_span else (hz, best)

    @staticmethod
    def clock_per_xcd(c0, c1):
        """Every XCD that was stamped before both readings: [{xcd, mhz, span_ms}] -- what `clock_hz_between` chose from (a short span reads noisier)."""
        out = []
        for x in range(8):
            if c0[x, 1] == 0 or c1[x, 1] <= c0[x, 1]:
                continue
            span = int(c1[x, 1]) - int(c0[x, 1])
            out.append({"xcd": x, "mhz": (int(c1[x, 0]) - int(c0[x, 0])) / span * 1e2, "span_ms": span / 1e5})
        return out

    def cost_stats(self, max_launches=1024):
        """(sum, max) over the envs of the wave cycles each bp_step launch took since `enable_timing(True)`: uint64 [launches, 2], in shader cycles."""
        out = np.zeros((max_launches, 2), np.uint64)
        n = C.c_int32()
        _lib.check(self.L, self.h, self.L.bp_get_cost_stats(self.h, out.ctypes.data_as(C.c_void_p), int(max_launches), C.byref(n)), "bp_get_cost_stats")
        return out[: n.value] << np.uint64(8)

    def sched_warnings(self):
        """(watchdog events, envs finished by the completion launch) of the step scheduler since load: (0, 0) unless a scheduler fault occurred."""
        out = np.zeros(2, np.int32)
        _lib.check(self.L, self.h, self.L.bp_sched_warnings(self.h, out.ctypes.data_as(C.c_void_p)), "bp_sched_warnings")
        return int(out[0]), int(out[1])

    def pair_stats(self):
        """Two environments per wavefront (bp_get_pair_stats): dict of the pairing mode and limits, and the cumulative counters since load."""
        out = np.zeros(16, np.int32)
        _lib.check(self.L, self.h, self.L.bp_get_pair_stats(self.h, out.ctypes.data_as(C.c_void_p)), "bp_get_pair_stats")
        keys = ["mode", "solo_first", "max_arbiter_lanes", "max_velocity_slots", "max_moving", "max_active", "max_warm_x_colours", "max_work_rate",
                "paired_first_tasks", "paired_tasks_from_queues", "envs_finished_in_a_pair", "envs_left_as_heavy", "heavy_envs_queued", "light_envs_queued"]
        return dict(zip(keys, out.tolist()))

    def sched_chunk(self):
        """Sub-steps per chunk of the preemptive step scheduler, 0 = one wavefront per env for the whole step."""
        return int(self.L.bp_sched_chunk(self.h))

    def set_cost_hint(self, costs):
        """Dispatch-order hint for the next step (uint32 per env, larger = earlier); by default the previous step's wave cycles."""
        c = np.ascontiguousarray(costs, dtype=np.uint32)
        assert c.shape == (self.num_envs,)
        _lib.check(self.L, self.h, self.L.bp_set_step_cost_hint(self.h, c.ctypes.data_as(C.c_void_p)), "bp_set_step_cost_hint")

    def enable_timing(self, on=True):
        self.L.bp_enable_timing(self.h, int(on))

    def kernel_time_ms(self):
        p, r, n = C.c_double(), C.c_double(), C.c_int32()
        _lib.check(self.L, self.h, self.L.bp_kernel_time_ms(self.h, C.byref(p), C.byref(r), C.byref(n)), "bp_kernel_time_ms")
        return p.value, r.value, n.value

    def debug_trace(self, buf, env=0):
        """Per-sub-step pose trace of one env into `buf`; only the -DBP_DEBUG_PATHS twin of the library records it (the product library returns BP_ESTATE
        for a non-null buffer, which surfaces here instead of leaving an all-zero trace behind)."""
        self._dbg = buf
        _lib.check(self.L, self.h, self.L.bp_debug_trace(self.h, _ptr(buf) if buf is not None else None, int(env)), "bp_debug_trace")


class ShipIceEnv(Env):
    """Reference-shaped single environment (E = 1) on the GPU path.

    Same surface as the reference ShipIceEnv (ship_ice_env.py:33-355): ``reset(seed, options) -> (obs, info)``,
    ``step(action) -> (obs, reward, terminated, False, info)`` with numpy observations and the reference's info keys.
    ``cfg.random_start`` (:201-203) re-draws the start x of every episode -- from a counter RNG keyed (start_seed, env, episode)
    instead of python's global ``random`` (``BatchedShipIceEnv.start_uniform`` reproduces the draw) -- and every reset then runs
    the 1000 settle sub-steps with the ship at that pose, like the reference.
    """

    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 4}

    def __init__(self, cfg=None, trials=None, device="cuda:0", num_trials=100, base_seed=0):
        super().__init__()
        self._b = BatchedShipIceEnv(1, cfg=cfg, trials=trials, device=device, num_trials=num_trials, base_seed=base_seed)
        self.cfg = self._b.cfg
        self.local_window_v_shift = 2
        self.beta = 30
        self.directional_reward_scale = 1.0
        self.episode_idx = None
        self.goal = (0, self.cfg.goal_y)
        self.path = None
        self.low_dim_state = self.cfg.low_dim_state
        self.max_yaw_rate_step = (np.pi / 2) / 7
        self.action_space = spaces.Box(low=-1, high=1, dtype=np.float32)
        self.env_max_trial = len(self._b.trials)
        if self.low_dim_state:
            n = len(self._b.trials[0]["obstacles"])
            self.observation_space = spaces.Box(low=-10, high=30, shape=(n * 2,), dtype=np.float64)
        else:
            if self.cfg.egocentric_obs:
                obs_shape = self._b.obs_shape
            else:  # planner observation (ship_ice_env.py:96-99)
                obs_shape = (2, int(self.cfg.occ.map_height / 0.2), int(self.cfg.occ.map_width / 0.2))
            self.observation_space = spaces.Box(low=0, high=255, shape=obs_shape, dtype=np.uint8)
        self.yaw_lim = (0, np.pi)
        self.boundary_violation_limit = 0.0
        self.total_work = [0, []]
        self.t = 0

    def _polys(self):
        verts, cnt = self._b.world_polys()
        verts = verts[0].cpu().numpy()
        cnt = cnt[0].cpu().numpy()
        nb = int(self._b.num_bodies()[0])
        return [verts[i, : cnt[i]].copy() for i in range(1, nb)]

    def _observation(self, obstacles):
        if self.low_dim_state:
            nb = int(self._b.num_bodies()[0])
            return self._b.low_dim_obs()[0, : nb - 1].reshape(-1).cpu().numpy()
        if not self.cfg.egocentric_obs:
            return self._b.observe_global()[0].cpu().numpy()
        return self._b.obs[0].cpu().numpy()

    def reset(self, seed=None, options=None):
        self.episode_idx = 0 if self.episode_idx is None else self.episode_idx + 1
        if self.episode_idx:
            self._b.check_errors()   # capacity flags of the episode that just ended (raises BpError)
        self._b.reset()
        self.t = 0
        self.total_work = [0, []]
        info_t = self._b.info[0].cpu().numpy()
        obstacles = self._polys()
        self.obstacles = obstacles
        info = {"state": (round(float(info_t[0]), 2), round(float(info_t[1]), 2), round(float(info_t[2]), 2)),
                "total_work": self.total_work[0], "obs": obstacles}
        return self._observation(obstacles), info

    def step(self, action):
        self.t += 1
        # the action keeps the caller's precision, like `action * self.max_yaw_rate_step` in the reference (ship_ice_env.py:265)
        a = torch.tensor([float(np.asarray(action, dtype=np.float64).reshape(-1)[0])], dtype=torch.float64)
        self._b.step(a)
        it = self._b.info[0].cpu().numpy()
        reward = float(self._b.reward[0].item())
        terminated = bool(self._b.terminated[0].item())
        obstacles = self._polys()
        self.obstacles = obstacles
        work = float(it[4])
        self.total_work[0] = float(it[3])
        self.total_work[1].append(work)
        info = {"state": (round(float(it[0]), 2), round(float(it[1]), 2), round(float(it[2]), 2)),
                "total_work": self.total_work[0], "collision reward": float(it[5]), "scaled collision reward": float(it[6]),
                "dist reward": float(it[7]), "trial_success": bool(it[8]), "obs": obstacles}
        if self.cfg.log_obs:
            self.log_observation()
        return self._observation(obstacles), reward, terminated, False, info

    def log_observation(self):
        """`cfg.log_obs` (ship_ice_env.py:352-353, 412-479): one PNG per observation channel under <cfg.output_dir>/t<episode_idx>/, named like the reference's
        files -- egocentric: <t>_con (occupancy), _orientation, _edt, _footprint; planner mode: <t>_con (5x5 block-mean occupancy), _footprint (obs_log.py)."""
        from ..obs_log import dump_channels
        if self.cfg.egocentric_obs:
            o = self._b.obs[0].cpu().numpy()           # channels [footprint, goal-line distance, orientation, occupancy] (ship_ice_env.py:392)
            ch = {"con": o[3], "orientation": o[2], "edt": o[1], "footprint": o[0]}
        else:
            o = self._b.observe_global()[0].cpu().numpy()   # [occupancy, footprint] (ship_ice_env.py:405)
            ch = {"con": o[0], "footprint": o[1]}
        return dump_channels(self.cfg.output_dir, self.episode_idx, self.t, ch)

    def update_path(self, new_path):
        self.path = new_path

    def render(self, mode="human", close=False):
        raise NotImplementedError("rendering (pygame) is outside the accelerated path")

    def close(self):
        self._b.close()
