"""Configuration objects for benchpush_amd.

``DotDict`` mirrors the reference's attribute-dict (benchpush/common/utils/utils.py:257-291): dot access,
``to_dict`` / ``to_dot_dict`` / ``load_from_file`` with the full yaml ``Loader`` (the reference configs use
``!!python/tuple``, ship_ice_nav/config.yaml:40-42).  ``merge_user_cfg`` restates the one-level-deep merge
every reference env performs in ``__init__`` (ship_ice_env.py:44-56).
"""
import math
import os

import yaml

try:  # same preference order as the reference (utils.py imports CLoader if present)
    from yaml import CLoader as _Loader
except ImportError:  # pragma: no cover
    from yaml import Loader as _Loader

_CFG_DIR = os.path.join(os.path.dirname(__file__), "configs")


class DotDict(dict):
    """dot.notation access to dictionary attributes (reference: utils.py:257-291)."""

    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__

    def __getattr__(self, attr):
        if attr not in self.keys():
            raise AttributeError
        return self.get(attr)

    @staticmethod
    def to_dict(d):
        return {k: DotDict.to_dict(d[k]) if type(v) is DotDict else v for k, v in d.items()}

    @staticmethod
    def to_dot_dict(d):
        return DotDict({k: DotDict.to_dot_dict(d[k]) if type(v) is dict else v for k, v in d.items()})

    @staticmethod
    def load_from_file(fp):
        with open(fp, "r") as fd:
            cfg = yaml.load(fd, Loader=_Loader)
        return DotDict.to_dot_dict(cfg)


def default_cfg(name):
    """Load the packaged default config for an env id family ('ship_ice')."""
    return DotDict.load_from_file(os.path.join(_CFG_DIR, name + ".yaml"))


def merge_user_cfg(base, cfg):
    """One-level-deep override, exactly as ship_ice_env.py:44-56."""
    if cfg is not None:
        for cfg_type in cfg:
            if type(cfg[cfg_type]) is DotDict or type(cfg[cfg_type]) is dict:
                if cfg_type not in base:
                    base[cfg_type] = DotDict()
                for param in cfg[cfg_type]:
                    base[cfg_type][param] = cfg[cfg_type][param]
            else:
                base[cfg_type] = cfg[cfg_type]
    return base


def ship_ice_physics_params(cfg):
    """Flatten a ship-ice cfg into the scalar physics/raster parameters shared by the C ABI.

    Chipmunk defaults that the reference never overrides (ship_ice_env.py:117-120 sets only iterations,
    gravity, damping): collision_slop 0.1, collision_bias (1-0.1)**60, collision_persistence 3.
    """
    dt_sub = cfg.dt / cfg.sim.steps
    collision_bias = math.pow(1.0 - 0.1, 60.0)
    return dict(
        dt=float(cfg.dt),
        steps=int(cfg.sim.steps),
        iterations=int(cfg.sim.iterations),
        persistence=3,
        settle_steps=1000,                       # ship_ice_env.py:218
        damping_pow=math.pow(float(cfg.sim.damping), dt_sub),
        bias_coef=1.0 - math.pow(collision_bias, dt_sub),
        slop=0.1,
        target_speed=float(cfg.target_speed),
        max_yaw_rate=(math.pi / 2) / 7,          # ship_ice_env.py:71
        map_w=float(cfg.occ.map_width),
        map_h=float(cfg.occ.map_height),
        goal_y=float(cfg.goal_y),
        m_to_pix=float(cfg.occ.m_to_pix_scale),
        density=float(cfg.sim.obstacle_density),
        poly_radius=0.02,                        # sim_utils.py:144, ship.py:90
        elasticity=0.01,
        friction=1.0,
        beta=30.0,                               # ship_ice_env.py:60
        boundary_penalty=-50.0,                  # ship_ice_env.py:30
        terminal_reward=200.0,                   # ship_ice_env.py:31
        local_w=6.0,
        local_h=6.0,                             # ship_ice_env.py:91
        vshift=2.0,                              # ship_ice_env.py:58
        obs_range=12.0,                          # ship_ice_env.py:383
        random_start=int(bool(cfg.get("random_start", False))),     # ship_ice_env.py:201-203
        start_x_range=float(cfg.get("start_x_range", 11.0) or 11.0),
        start_seed=int(cfg.get("start_seed", 0) or 0),              # key of the counter RNG that replaces python's global `random`
        ship_mass=float(cfg.ship.mass),                             # ShipIceMetric(ship_mass=env.cfg.ship.mass)
    )


def maze_physics_params(cfg):
    """Flatten a maze-NAMO cfg (env1/env2 already selected into cfg.env, maze_NAMO_env.py:68-73) into C-ABI scalars."""
    dt_sub = cfg.dt / cfg.sim.steps
    collision_bias = math.pow(1.0 - 0.1, 60.0)
    return dict(
        dt=float(cfg.dt), steps=int(cfg.sim.steps), iterations=int(cfg.sim.iterations), persistence=3, settle_steps=1000,
        damping_pow=math.pow(float(cfg.sim.damping), dt_sub), bias_coef=1.0 - math.pow(collision_bias, dt_sub), slop=0.1,
        target_speed=float(cfg.target_speed),
        max_yaw_rate=(math.pi / 2) / 15,          # maze_NAMO_env.py:101
        map_w=float(cfg.env.width), map_h=float(cfg.env.length),
        goal_y=float(cfg.env.goal_y), goal_x=float(cfg.env.goal_x),
        goal_reach=float(cfg.goal_radius + cfg.robot.min_r),   # maze_NAMO_env.py:533
        m_to_pix=float(cfg.occ.m_to_pix_scale), density=float(cfg.sim.obstacle_density), poly_radius=0.02,
        elasticity=0.01, friction=1.0,
        beta=1.5, k_increment=150.0,              # maze_NAMO_env.py:80-82
        boundary_penalty=-50.0, terminal_reward=200.0,
        local_w=float(cfg.occ.local_width), local_h=float(cfg.occ.local_height), vshift=0.0, obs_range=0.0,
        wall_radius=0.5,                          # sim_utils.py:177
        ship_mass=float(cfg.robot.mass),          # m0 of MazeNamoMetric.compute_effort_score (maze_namo_metric.py:36-42)
    )


def maze_walls(cfg):
    """construct_maze_walls (maze_NAMO_env.py:357-375) as [[ax, ay, bx, by], ...]."""
    L, W = cfg.env.length, cfg.env.width
    if cfg.maze_version == 1:
        w = [[(0, 0), (W, 0)], [(0, 0), (0, L)], [(W, 0), (W, L)], [(0, L), (W, L)], [(2 * W / 2, L), (2 * W / 2, 5)],
             [(W / 2, 0), (W / 2, L - L / 3)]]
    elif cfg.maze_version == 2:
        w = [[(0, 0), (W, 0)], [(0, 0), (0, L)], [(W, 0), (W, L)], [(0, L), (W, L)], [(W / 3, 0), (W / 3, 2 * L / 3)],
             [(2 * W / 3, L), (2 * W / 3, L / 3)]]
    else:
        raise Exception("Invalid Maze Version!")
    return [[float(a[0]), float(a[1]), float(b[0]), float(b[1])] for a, b in w]
