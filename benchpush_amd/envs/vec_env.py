"""Stable-Baselines3-shaped vectorised environment over a batched GPU env.

The reference trains with ``PPO("CnnPolicy", env, ...)`` on a single ``gym.Env`` (baselines/ship_ice_nav/ppo/policy.py:29-69, ResNet18 extractor
baselines/feature_extractors.py:11-45), which SB3 wraps into a ``DummyVecEnv`` of one environment.  ``BatchedVecEnv`` exposes all E device environments
through the same ``VecEnv`` protocol instead (``reset()``, ``step_async(actions)`` / ``step_wait()`` -> ``(obs, rewards, dones, infos)``, auto-reset with
``infos[i]['terminal_observation']`` and ``infos[i]['TimeLimit.truncated']``).  If stable_baselines3 is importable the class derives from its ``VecEnv``;
this image ships without it, so the protocol is implemented directly.

What the adapter costs per step is what bounds a learner behind it, so it is built to cost nothing it does not have to (tools/bench_vecenv.py measures it):

* ``to_numpy=False`` (on-device learners): observations, rewards and dones stay device tensors and ``step_wait`` never synchronises with the GPU -- the
  done mask goes to the masked reset as a device tensor, the observations of the envs about to be reset are saved on the device by
  ``bp_copy_rows_masked`` (cost proportional to the rows that finished), the step counters are updated with tensor ops, and capacity flags are polled
  every ``check_every`` steps instead of on every batch with a finished episode.
* ``to_numpy=True`` (SB3 itself): one non-blocking device-to-host copy per output into pre-allocated PINNED host buffers and ONE synchronisation per step.
  SB3's ``DummyVecEnv`` hands out a deep copy of its observation buffer, and its learners rely on that: ``collect_rollouts`` calls ``env.step()`` first and only
  then stores ``self._last_obs`` -- the batch returned by the PREVIOUS step -- in the rollout / replay buffer.  A batch must therefore stay intact while the next
  step is taken: the observation batch rotates through ``obs_buffers`` (default 2) pinned buffers, so the array returned by step t is rewritten by step
  t + ``obs_buffers``, not before (``obs_buffers=0`` returns a fresh copy each step: no lifetime rule at all, one more host pass over the batch).  Rewards,
  dones and the info block are small and always returned as copies.
* ``infos`` is a lazy sequence in both modes: ``infos[i]`` builds env i's dict on first access from one host copy of the info block; only ``len``,
  indexing and iteration are offered, which is all SB3 uses.  4096 dicts are no longer built per step.
"""
import collections.abc
import ctypes as C

import numpy as np
import torch

from ..gym_shim import spaces

try:  # pragma: no cover - not installed in the build image
    from stable_baselines3.common.vec_env import VecEnv as _Base
except ImportError:
    _Base = object

__all__ = ["BatchedVecEnv", "LazyInfos", "make_ship_ice_vec_env", "make_maze_vec_env", "make_box_delivery_vec_env", "make_area_clearing_vec_env"]


class LazyInfos(collections.abc.Sequence):
    """``infos`` of one ``step_wait``: a sequence of E dicts built on demand.

    ``infos[i]`` holds the env's info scalars under the adapter's ``info_keys`` and, for an env whose episode ended in this step,
    ``'terminal_observation'`` (numpy, or a device tensor with ``to_numpy=False``) and ``'TimeLimit.truncated'``.  The first access copies the
    [E, K] info block (and the done / truncation masks) to the host once; nothing is copied if nobody looks.

    Lifetime: the info / done / truncation arrays handed to the constructor are private snapshots of the step (never rewritten).  The terminal observations
    live in one of the adapter's two rotating device buffers: ``infos[i]`` must be read (materialised) before the step after the next one rewrites that
    buffer -- SB3 reads them right after ``step_wait`` -- and in numpy mode the dict then holds its own host copy."""

    def __init__(self, keys, info, done, trunc, term_obs, to_numpy):
        self._keys, self._info, self._done, self._trunc, self._term_obs, self._to_numpy = keys, info, done, trunc, term_obs, to_numpy
        self._host = None
        self._cache = {}

    def _pull(self):
        if self._host is None:
            as_np = lambda t: t if isinstance(t, np.ndarray) else t.cpu().numpy()   # noqa: E731
            self._host = (as_np(self._info), as_np(self._done).astype(bool), as_np(self._trunc).astype(bool))
        return self._host

    def __len__(self):
        return int(self._info.shape[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        i = int(i)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        d = self._cache.get(i)
        if d is None:
            info_h, done_h, trunc_h = self._pull()
            d = dict(zip(self._keys, info_h[i].tolist()))
            if done_h[i]:
                to = self._term_obs[i]
                d["terminal_observation"] = to.cpu().numpy() if (self._to_numpy and torch.is_tensor(to)) else to
                d["TimeLimit.truncated"] = bool(trunc_h[i])
            self._cache[i] = d
        return d

    def done_indices(self):
        """Host indices of the envs whose episode ended in this step (one small copy)."""
        return np.nonzero(self._pull()[1])[0]


class BatchedVecEnv(_Base):
    def __init__(self, batched_env, info_keys, max_episode_steps=None, to_numpy=True, action_shape=(), check_every=64, obs_buffers=2):
        self.env = batched_env
        self.num_envs = batched_env.num_envs
        self.observation_space = spaces.Box(low=0, high=255, shape=batched_env.obs_shape, dtype=np.uint8)
        self.action_space = spaces.Box(low=-1, high=1, shape=action_shape, dtype=np.float32)
        self._adim = int(np.prod(action_shape)) if len(action_shape) else 1
        self.info_keys = list(info_keys)
        self.max_episode_steps = max_episode_steps
        self.to_numpy = to_numpy
        self.check_every = int(check_every)
        dv = batched_env.device
        self._steps = torch.zeros(self.num_envs, dtype=torch.int64, device=dv)
        self._actions = None
        self._nstep = 0
        # rows of finished envs, saved before the reset; two buffers in rotation so that the infos of step t survive step t + 1
        self._term_obs_ring = [torch.zeros((self.num_envs,) + tuple(batched_env.obs_shape), dtype=torch.uint8, device=dv) for _ in range(2)]
        self._term_obs = self._term_obs_ring[0]
        self.obs_buffers = int(obs_buffers)
        self._obs_turn = 0
        self._host = None          # pinned host buffers of the to_numpy path, allocated on first use
        if _Base is not object:
            _Base.__init__(self, self.num_envs, self.observation_space, self.action_space)

    # -- host buffers of the numpy path: pinned, allocated once ------------------------------------------
    def _host_buffers(self):
        if self._host is None:
            E, e = self.num_envs, self.env
            pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)   # noqa: E731
            self._host = {"obs": [pin((E,) + tuple(e.obs_shape), torch.uint8) for _ in range(max(1, self.obs_buffers))],
                          "rew": pin((E,), torch.float64), "done": pin((E,), torch.bool),
                          "trunc": pin((E,), torch.bool), "info": pin(tuple(e.info.shape), torch.float64)}
        return self._host

    def _next_obs_buffer(self):
        """The pinned observation buffer of this step: the one whose batch was handed out longest ago."""
        ring = self._host_buffers()["obs"]
        buf = ring[self._obs_turn % len(ring)]
        self._obs_turn += 1
        return buf

    def _obs_out(self, buf):
        return buf.numpy().copy() if self.obs_buffers <= 0 else buf.numpy()

    def _save_terminal_rows(self, done_u8, obs):
        from .. import _lib
        e = self.env
        row_bytes = int(np.prod(e.obs_shape))
        stream = C.c_void_p(torch.cuda.current_stream(e.device).cuda_stream)
        _lib.check(e.L, e.h, e.L.bp_copy_rows_masked(e.h, C.c_void_p(done_u8.data_ptr()), C.c_void_p(obs.data_ptr()), C.c_void_p(self._term_obs.data_ptr()),
                                                     self.num_envs, row_bytes, stream), "bp_copy_rows_masked")

    def reset(self):
        obs, _ = self.env.reset()
        self._steps.zero_()
        if not self.to_numpy:
            return obs
        buf = self._next_obs_buffer()
        buf.copy_(obs, non_blocking=True)
        torch.cuda.current_stream(self.env.device).synchronize()
        return self._obs_out(buf)

    def step_async(self, actions):
        a = torch.as_tensor(np.asarray(actions, dtype=np.float32).reshape(self.num_envs, self._adim)) if not torch.is_tensor(actions) else actions
        self._actions = a.reshape(self.num_envs * self._adim).to(torch.float32).to(torch.float64)

    def step_wait(self):
        env = self.env
        obs, rew, term, trunc, info = env.step(self._actions)
        term_b, env_trunc = term.bool(), trunc.bool()
        self._steps += 1
        trunc_b = env_trunc & ~term_b                                   # truncation reported by the env itself (box-delivery, area-clearing)
        if self.max_episode_steps is not None:                          # gym's TimeLimit (ids registered with max_episode_steps)
            trunc_b = trunc_b | ((self._steps >= self.max_episode_steps) & ~term_b)
        done = term_b | env_trunc | trunc_b
        done_u8 = done.to(torch.uint8)
        rew_out = rew.clone()                                           # the env's buffers are rewritten by the next step
        info_out = info.clone()
        self._term_obs = self._term_obs_ring[self._nstep & 1]
        self._save_terminal_rows(done_u8, obs)                          # terminal observations, before the auto-reset overwrites those rows
        obs, _ = env.reset(done_u8)                                     # masked reset, device mask: a batch with no finished env launches two empty kernels
        self._steps.mul_((~done).to(torch.int64))
        self._nstep += 1
        if self.check_every > 0 and self._nstep % self.check_every == 0:
            # an in-kernel capacity overflow (neighbour list, arbiter / velocity slots, colours) only raises a per-env flag while the step carries on
            # with dropped items: surfaced here every `check_every` steps (a host synchronisation) and in close()
            env.check_errors()
        if not self.to_numpy:
            return obs, rew_out, done, LazyInfos(self.info_keys, info_out, done, trunc_b, self._term_obs, False)
        hb = self._host_buffers()
        obuf = self._next_obs_buffer()
        obuf.copy_(obs, non_blocking=True); hb["rew"].copy_(rew_out, non_blocking=True); hb["done"].copy_(done, non_blocking=True)
        hb["trunc"].copy_(trunc_b, non_blocking=True); hb["info"].copy_(info_out, non_blocking=True)
        torch.cuda.current_stream(env.device).synchronize()             # the one synchronisation of the step
        # rewards, dones and the info block are small: returned as copies (SB3's DummyVecEnv copies its buffers too); the observation batch is a view of
        # the pinned buffer of this turn, which the NEXT step does not touch (see the module docstring)
        done_h = hb["done"].numpy().copy()
        return (self._obs_out(obuf), hb["rew"].numpy().copy(), done_h,
                LazyInfos(self.info_keys, hb["info"].numpy().copy(), done_h, hb["trunc"].numpy().copy(), self._term_obs, True))

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        if getattr(self.env, "h", None):
            try:
                self.env.check_errors()
            finally:
                self.env.close()

    # minimal VecEnv plumbing used by SB3
    def seed(self, seed=None):
        return [None] * self.num_envs

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.env, attr_name)] * self.num_envs

    def set_attr(self, attr_name, value, indices=None):
        setattr(self.env, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        return [getattr(self.env, method_name)(*method_args, **method_kwargs)] * self.num_envs

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.num_envs


def make_ship_ice_vec_env(num_envs, cfg=None, to_numpy=True, **kw):
    from .. import _lib
    from .ship_ice import BatchedShipIceEnv
    return BatchedVecEnv(BatchedShipIceEnv(num_envs, cfg=cfg, **kw), _lib.INFO_KEYS, max_episode_steps=300, to_numpy=to_numpy)


def make_maze_vec_env(num_envs, cfg=None, to_numpy=True, **kw):
    from .maze_namo import MAZE_INFO_KEYS, BatchedMazeEnv
    return BatchedVecEnv(BatchedMazeEnv(num_envs, cfg=cfg, **kw), MAZE_INFO_KEYS, max_episode_steps=400, to_numpy=to_numpy)


def make_box_delivery_vec_env(num_envs, cfg=None, to_numpy=True, **kw):
    from .box_delivery import BD_INFO_KEYS, BatchedBoxDeliveryEnv
    env = BatchedBoxDeliveryEnv(num_envs, cfg=cfg, **kw)
    return BatchedVecEnv(env, BD_INFO_KEYS, max_episode_steps=30000, action_shape=(env.action_dim,), to_numpy=to_numpy)


def make_area_clearing_vec_env(num_envs, cfg=None, to_numpy=True, **kw):
    from .area_clearing import AC_INFO_KEYS, BatchedAreaClearingEnv
    env = BatchedAreaClearingEnv(num_envs, cfg=cfg, **kw)
    return BatchedVecEnv(env, AC_INFO_KEYS, max_episode_steps=30000, action_shape=(env.action_dim,), to_numpy=to_numpy)
