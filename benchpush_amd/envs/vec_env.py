"""Stable-Baselines3-shaped vectorised environment over a batched GPU env.

The reference trains with ``PPO("CnnPolicy", env, ...)`` on a single ``gym.Env`` (baselines/ship_ice_nav/ppo/policy.py:29-69),
which SB3 wraps into a ``DummyVecEnv`` of one environment.  ``BatchedVecEnv`` exposes all E device environments through the
same ``VecEnv`` protocol instead (``reset()``, ``step_async(actions)`` / ``step_wait()`` -> ``(obs, rewards, dones, infos)``,
auto-reset with ``infos[i]['terminal_observation']`` and ``infos[i]['TimeLimit.truncated']``).  If stable_baselines3 is
importable the class derives from its ``VecEnv``; this image ships without it, so the protocol is implemented directly.
Observations stay on the GPU until ``step_wait`` converts the batch once (``to_numpy=False`` keeps torch tensors).
"""
import numpy as np
import torch

from ..gym_shim import spaces

try:  # pragma: no cover - not installed in the build image
    from stable_baselines3.common.vec_env import VecEnv as _Base
except ImportError:
    _Base = object

__all__ = ["BatchedVecEnv", "make_ship_ice_vec_env", "make_maze_vec_env", "make_box_delivery_vec_env", "make_area_clearing_vec_env"]


class BatchedVecEnv(_Base):
    def __init__(self, batched_env, info_keys, max_episode_steps=None, to_numpy=True, action_shape=()):
        self.env = batched_env
        self.num_envs = batched_env.num_envs
        self.observation_space = spaces.Box(low=0, high=255, shape=batched_env.obs_shape, dtype=np.uint8)
        self.action_space = spaces.Box(low=-1, high=1, shape=action_shape, dtype=np.float32)
        self._adim = int(np.prod(action_shape)) if len(action_shape) else 1
        self.info_keys = list(info_keys)
        self.max_episode_steps = max_episode_steps
        self.to_numpy = to_numpy
        self._steps = torch.zeros(self.num_envs, dtype=torch.int64, device=batched_env.device)
        self._actions = None
        if _Base is not object:
            _Base.__init__(self, self.num_envs, self.observation_space, self.action_space)

    def _out(self, t):
        return t.cpu().numpy() if self.to_numpy else t

    def reset(self):
        obs, _ = self.env.reset()
        self._steps.zero_()
        return self._out(obs)

    def step_async(self, actions):
        a = torch.as_tensor(np.asarray(actions, dtype=np.float32).reshape(self.num_envs, self._adim)) if not torch.is_tensor(actions) else actions
        self._actions = a.reshape(self.num_envs * self._adim).to(torch.float32).to(torch.float64)

    def step_wait(self):
        obs, rew, term, trunc, info = self.env.step(self._actions)
        self._steps += 1
        trunc_b = torch.zeros_like(term, dtype=torch.bool)
        if self.max_episode_steps is not None:  # gym's TimeLimit (ids registered with max_episode_steps)
            trunc_b = (self._steps >= self.max_episode_steps) & ~term.bool()
        trunc_b = trunc_b | (trunc.bool() & ~term.bool())   # truncation reported by the env itself (box-delivery, area-clearing)
        done = term.bool() | trunc.bool() | trunc_b
        info_h = info.cpu().numpy()
        done_h = done.cpu().numpy()
        trunc_h = trunc_b.cpu().numpy()
        infos = [dict(zip(self.info_keys, info_h[e].tolist())) for e in range(self.num_envs)]
        rew_out = self._out(rew.clone())
        if done_h.any():
            term_obs = obs[done].cpu().numpy()  # terminal observations before the auto-reset overwrites them
            for k, e in enumerate(np.nonzero(done_h)[0]):
                infos[e]["terminal_observation"] = term_obs[k]
                infos[e]["TimeLimit.truncated"] = bool(trunc_h[e])
            # an in-kernel capacity overflow (neighbour list, arbiter / velocity slots, colours) only raises a per-env flag while the
            # step carries on with dropped items: surface it here, once per batch of finished episodes, instead of never
            self.env.check_errors()
            obs, _ = self.env.reset(done)
            self._steps[done] = 0
        return self._out(obs), rew_out, (done_h if self.to_numpy else done), infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
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


def make_ship_ice_vec_env(num_envs, cfg=None, **kw):
    from .. import _lib
    from .ship_ice import BatchedShipIceEnv
    return BatchedVecEnv(BatchedShipIceEnv(num_envs, cfg=cfg, **kw), _lib.INFO_KEYS, max_episode_steps=300)


def make_maze_vec_env(num_envs, cfg=None, **kw):
    from .maze_namo import MAZE_INFO_KEYS, BatchedMazeEnv
    return BatchedVecEnv(BatchedMazeEnv(num_envs, cfg=cfg, **kw), MAZE_INFO_KEYS, max_episode_steps=400)


def make_box_delivery_vec_env(num_envs, cfg=None, **kw):
    from .box_delivery import BD_INFO_KEYS, BatchedBoxDeliveryEnv
    env = BatchedBoxDeliveryEnv(num_envs, cfg=cfg, **kw)
    return BatchedVecEnv(env, BD_INFO_KEYS, max_episode_steps=30000, action_shape=(env.action_dim,))


def make_area_clearing_vec_env(num_envs, cfg=None, **kw):
    from .area_clearing import AC_INFO_KEYS, BatchedAreaClearingEnv
    env = BatchedAreaClearingEnv(num_envs, cfg=cfg, **kw)
    return BatchedVecEnv(env, AC_INFO_KEYS, max_episode_steps=30000, action_shape=(env.action_dim,))
