"""ShipIceEnv.step's control / yaw / boundary / reward / termination logic pinned against golden vectors produced by the reference
class itself running on a stand-in space (tests/golden/make_golden_step_logic.py): the oracle runs its full restated physics with
the single floe parked far from the ship, so both integrate the same kinematic motion."""
import json
import os

import numpy as np

from benchpush_amd.config import default_cfg, ship_ice_physics_params
from oracle.oracle import OracleShipIce

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "step_logic_golden.json")))


def test_ship_ice_step_logic_matches_reference_class():
    cfg = default_cfg("ship_ice")
    floe = np.array(G["floe"])
    trial = {"goal": (0, cfg.goal_y), "ship_state": (6, 1, np.pi / 2),
             "obstacles": [{"vertices": floe, "centre": floe.mean(0), "radius": 0.7}]}
    for case in G["ship_ice"]:
        o = OracleShipIce(ship_ice_physics_params(cfg), cfg.ship.vertices, cfg.ship.head, cfg.ship.tail)
        o.reset(trial, start=case["start"], observe=False)
        for row in case["steps"]:
            _, r, term, info = o.step(row["action"], observe=False)
            assert term == row["terminated"]
            assert abs(r - row["reward"]) < 1e-12, (case["start"], r, row["reward"])
            pose = np.array([info["x"], info["y"], info["theta"]])
            assert np.abs(pose - np.array(row["pose"])).max() < 1e-12
            assert [round(float(v), 2) for v in pose] == row["state"]
            assert bool(info["trial_success"]) == row["trial_success"] and abs(info["dist_reward"] - row["dist_reward"]) < 1e-15
            assert info["total_work"] == row["total_work"] == 0.0


def test_maze_step_logic_matches_reference_class():
    """MazeNAMO.step (maze_NAMO_env.py:402-485): the same injected goal map on both sides, the box far from the robot's track."""
    from benchpush_amd.config import maze_physics_params, maze_walls
    from benchpush_amd.envs.maze_namo import _maze_cfg
    from oracle.oracle import OracleMaze
    cfg = _maze_cfg(None)
    params = maze_physics_params(cfg)
    H, W = int(cfg.env.length * cfg.occ.m_to_pix_scale), int(cfg.env.width * cfg.occ.m_to_pix_scale)
    i, j = np.indices((H, W))
    dmap = ((i * 37 + j * 91) % 1000) / 1000.0
    for case in G["maze"]:
        o = OracleMaze(params, cfg.robot.vertices, cfg.robot.wheel_vertices, cfg.obstacle_size)
        o.reset({"centres": [case["box"]], "walls": maze_walls(cfg), "start": case["start"]}, observe=False)
        o.set_dist_map(dmap)
        for row in case["steps"]:
            _, r, term, info = o.step(row["action"], observe=False)
            assert term == row["terminated"] and info["wall_collision"] == 0
            assert abs(r - row["reward"]) < 1e-9, (case["start"], r, row["reward"])
            assert np.abs(np.array([info["x"], info["y"], info["theta"]]) - np.array(row["pose"])).max() < 1e-12
            assert abs(info["dist_increment_reward"] - row["dist_increment"]) < 1e-9 and bool(info["trial_success"]) == row["trial_success"]
            assert info["total_work"] == row["total_work"] == 0.0
