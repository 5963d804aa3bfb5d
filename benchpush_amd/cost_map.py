"""Planner cost map with the reference's CostMap surface (benchpush/common/cost_map.py:18-126), computed on the GPU from the
environment's current obstacles.

The reference planners build ``CostMap(horizon=..., ship_mass=..., **cfg.costmap)`` and call ``update(info['obs'], ship_pos_y, vs)``
every planning step (planners/lattice.py:37-38,78-79), then read ``cost_map``.  Here the map is produced by ``bp_costmap_update`` from
the device state of a ``ShipIceEnv`` / ``BatchedShipIceEnv`` -- ``info['obs']`` *is* that state -- so the polygons never visit the host.
"""
import numpy as np

MAX_COST = 1e10


class CostMap:
    def __init__(self, scale, m, n, alpha=10, ship_mass=1, horizon=None, margin=1, env=None):
        if env is None:
            raise ValueError("CostMap needs the environment whose obstacles it rasterises: CostMap(..., env=env)")
        self._b = getattr(getattr(env, "unwrapped", env), "_b", env)   # gym adapter -> its batched env
        self.scale, self.m, self.n = scale, m, n
        self.alpha, self.ship_mass, self.margin = alpha, ship_mass, margin
        self.horizon = horizon * scale if horizon else None            # cost_map.py:42
        self._horizon_m = horizon
        self.cost_maps = None
        self.cost_map = np.zeros((int(m * scale), int(n * scale)))
        self.boundary_cost()

    @property
    def shape(self):
        return self.cost_map.shape

    def boundary_cost(self):
        if not self.margin:
            return
        self.cost_map[:, :self.margin] = MAX_COST
        self.cost_map[:, -self.margin:] = MAX_COST

    def update(self, obstacles=None, ship_pos_y=0, vs=1):
        """``obstacles`` is accepted for signature compatibility; the environment's device state is what gets rasterised."""
        E = self._b.num_envs
        spy = np.broadcast_to(np.asarray(ship_pos_y, np.float64), (E,)).copy()
        self.cost_maps = self._b.cost_maps(self.scale, self.m, self.n, self.alpha, self.ship_mass, self._horizon_m, self.margin, spy, vs)
        self.cost_map = self.cost_maps[0].cpu().numpy()
