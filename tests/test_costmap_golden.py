"""Oracle cost maps vs. goldens produced by the reference's own CostMap class (tests/golden/make_golden_costmap.py;
benchpush/common/cost_map.py:27-126,284-287).  The set of costed cells must be identical; values within 1e-10 relative (the reference's
`** 0.5` / `** 2` / np.dot are evaluated here as sqrt / products / sequential sums, and (r^2 - d^2)/r^2 amplifies the last-place
differences near a floe's rim)."""
import json
import os

import numpy as np
import pytest

from benchpush_amd.config import default_cfg, ship_ice_physics_params
from benchpush_amd.envs.ship_ice import default_trials
from oracle import oracle as orc

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-10


def load_golden():
    with open(os.path.join(HERE, "costmap_golden.json")) as f:
        return np.load(os.path.join(HERE, "costmap_golden.npz")), json.load(f)


def oracle_at(case_actions, trial):
    cfg = default_cfg("ship_ice")
    cfg.concentration = 0.3
    trials = default_trials(0.3, 2, base_seed=21)
    o = orc.OracleShipIce(ship_ice_physics_params(cfg), cfg.ship.vertices, cfg.ship.head, cfg.ship.tail)
    o.reset(trials[trial], observe=False)
    for a in case_actions:
        o.step(a, observe=False)
    return o


def test_oracle_costmap_matches_reference_class():
    G, M = load_golden()
    cache = {}
    for c in M:
        if c["case"] not in cache:
            cache[c["case"]] = oracle_at(c["actions"], c["trial"])
        got = cache[c["case"]].costmap(c["scale"], c["m"], c["n"], c["alpha"], c["ship_mass"], c["horizon"], c["margin"], c["ship_pos_y"], c["vs"])
        ref = G["c%d_k%d" % (c["case"], c["cfg"])]
        assert got.shape == ref.shape
        assert np.array_equal(got != 0, ref != 0), (c["case"], c["cfg"])
        assert np.allclose(got, ref, rtol=RTOL, atol=0.0), (c["case"], c["cfg"])
        if c["margin"]:
            assert np.all(got[:, : c["margin"]] == 1e10) and np.all(got[:, -c["margin"]:] == 1e10)
        assert (ref[:, c["margin"]: ref.shape[1] - c["margin"]] > 0).sum() > 50     # the case is not vacuous
