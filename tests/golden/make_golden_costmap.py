"""Golden planner cost maps from the reference's own CostMap class (common/cost_map.py) on obstacle polygons exported from the oracle
(run ONLY in the build container, after `make -C oracle`):

    PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden_costmap.py

skimage.draw.polygon is absent: the reference's code calls this repository's restatement instead (oracle hook).  Data only.
"""
import json
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

from benchpush_amd.config import default_cfg, ship_ice_physics_params
from benchpush_amd.envs.ship_ice import default_trials
from oracle import oracle as orc

for m in ["shapely", "shapely.geometry", "pymunk"]:
    sys.modules[m] = MagicMock()
draw = types.ModuleType("skimage.draw")
draw.polygon = lambda r, c, shape=None: orc.draw_polygon(np.asarray(r, np.float64), np.asarray(c, np.float64), shape)
sk = types.ModuleType("skimage"); sk.draw = draw
sys.modules.update({"skimage": sk, "skimage.draw": draw})

from benchpush.common.cost_map import CostMap  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
cfg = default_cfg("ship_ice")
cfg.concentration = 0.3
trials = default_trials(0.3, 2, base_seed=21)
rng = np.random.RandomState(8)
cases, arrays = [], {}
# (scale, m, n, alpha, ship_mass, horizon, margin, vs): the lattice planner's configuration (lattice_config.yaml:40-44, lattice.py:37-38,78-79)
# and variations that exercise the horizon cull, a wide margin and a finer grid
CFGS = [(5, 76, 12, 10, 1, None, 1, 0.3 * 5 + 1e-8), (5, 40, 12, 10, 1, 8, 1, 1.0), (8, 40, 12, 2.5, 3.0, None, 2, 0.7), (5, 40, 12, 10, 1, 3, 0, 2.0)]
for ci, nsteps in enumerate([0, 6, 13]):
    o = orc.OracleShipIce(ship_ice_physics_params(cfg), cfg.ship.vertices, cfg.ship.head, cfg.ship.tail)
    o.reset(trials[ci % 2], observe=False)
    actions = [float(np.float32(a)) for a in rng.uniform(-1, 1, nsteps)]
    for a in actions:
        o.step(a, observe=False)
    polys, cnt = o.world_polys()
    obstacles = [polys[i, : cnt[i]].copy() for i in range(1, len(cnt))]
    ship_y = float(o.bodies()[0][1])
    for ki, (scale, m, n, alpha, mass, horizon, margin, vs) in enumerate(CFGS):
        cm = CostMap(scale=scale, m=m, n=n, alpha=alpha, ship_mass=mass, horizon=horizon, margin=margin)
        spy = ship_y * scale - 1.0
        cm.update(obstacles, spy, vs=vs)
        arrays["c%d_k%d" % (ci, ki)] = cm.cost_map.copy()
        cases.append({"case": ci, "cfg": ki, "trial": ci % 2, "actions": actions, "scale": scale, "m": m, "n": n, "alpha": alpha, "ship_mass": mass,
                      "horizon": horizon, "margin": margin, "vs": vs, "ship_pos_y": spy, "num_obstacles": len(cm.obstacles)})
np.savez_compressed(os.path.join(HERE, "costmap_golden.npz"), **arrays)
with open(os.path.join(HERE, "costmap_golden.json"), "w") as f:
    json.dump(cases, f)
print("wrote", len(cases), "cost maps;", [(c["case"], c["cfg"], c["num_obstacles"]) for c in cases])
