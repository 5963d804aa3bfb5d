"""Oracle observations vs. goldens produced by the reference's own ShipIceEnv / MazeNAMO.generate_observation + OccupancyGrid code
(tests/golden/make_golden_obs_pipeline.py; ship_ice_env.py:470-560, maze_NAMO_env.py:488-560, occupancy_map.py).  uint8, bit-exact."""
import json
import os

import numpy as np
import pytest

from benchpush_amd.config import default_cfg, maze_physics_params, maze_walls, ship_ice_physics_params
from benchpush_amd.envs.maze_namo import _maze_cfg
from benchpush_amd.envs.ship_ice import default_trials
from benchpush_amd.maze_scenario import generate_layout
from oracle import oracle as orc

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "obs_pipeline_golden.json")) as f:
        meta = json.load(f)
    return np.load(os.path.join(HERE, "obs_pipeline_golden.npz")), meta


def test_ship_ice_observations_match_reference_pipeline(golden):
    G, M = golden
    cfg = default_cfg("ship_ice")
    cfg.concentration = 0.3
    trials = default_trials(0.3, 2, base_seed=11)
    for ci, m in enumerate(M["ship_ice"]):
        o = orc.OracleShipIce(ship_ice_physics_params(cfg), cfg.ship.vertices, cfg.ship.head, cfg.ship.tail)
        obs, _ = o.reset(trials[m["trial"]])
        for a in m["actions"]:
            obs = o.step(a)[0]
        ref = G["ship%d_ego" % ci]
        assert ref.dtype == np.uint8 and ref.shape == obs.shape
        assert np.array_equal(obs, ref), "egocentric observation, case %d" % ci
        assert np.array_equal(o.observe_global(), G["ship%d_global" % ci]), "planner observation, case %d" % ci
        assert ref[0].max() > 0 and ref[1].max() > 0       # the case is not vacuous


def test_maze_observations_and_goal_map_match_reference_pipeline(golden):
    G, M = golden
    mcfg = _maze_cfg({"num_obstacles": 8})
    walls = maze_walls(mcfg)
    for ci, m in enumerate(M["maze"]):
        om = orc.OracleMaze(maze_physics_params(mcfg), mcfg.robot.vertices, mcfg.robot.wheel_vertices, mcfg.obstacle_size)
        obs = om.reset(generate_layout(mcfg, walls, m["seed"]))
        for a in m["actions"]:
            obs = om.step(a)[0]
        assert np.array_equal(obs, G["maze%d_obs" % ci]), "maze observation, case %d" % ci
        if ci == 0:
            norm, raw, _ = om.maps()
            assert np.array_equal(norm, G["maze_goal_map"])
            assert np.array_equal(raw, G["maze_goal_raw"])
