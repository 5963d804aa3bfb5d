"""Stated tolerance against the real reference (pymunk is absent): the spread of the results over Gauss-Seidel sweep orders.

tests/golden/order_envelope.json is the committed output of tools/order_envelope.py (100 trials x 300 steps x 5 orders, 30 %).  Here a
sample is re-run and held against it: what the solver order cannot touch must be exactly equal (the kinematic ship's pose, hence the
termination step and the success flag), what it can touch must stay inside the published envelope."""
import json
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from benchpush_amd.config import default_cfg, merge_user_cfg, ship_ice_physics_params
from benchpush_amd.envs.ship_ice import default_trials
from oracle.oracle import OracleShipIce

ENV = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "order_envelope.json")))


def _run(args):
    trial, tidx, steps, mode = args
    cfg = merge_user_cfg(default_cfg("ship_ice"), {"concentration": 0.3})
    o = OracleShipIce(ship_ice_physics_params(cfg), cfg.ship.vertices, cfg.ship.head, cfg.ship.tail)
    o.set_solve_order(mode, 1234 + tidx)
    o.reset(trial, observe=False)
    rng = np.random.default_rng(1000 + tidx)
    rows, snaps = [], {}
    for t in range(steps):
        _, r, term, info = o.step(float(rng.uniform(-1, 1)), observe=False)
        rows.append((info["x"], info["y"], info["theta"], float(term), info["trial_success"], info["total_work"], r))
        if t + 1 in (1, 2, 5, 10, 20):
            snaps[t + 1] = o.bodies()[1:, :2].copy()
        if term:
            break
    return np.array(rows), snaps


def test_published_envelope_is_what_the_tool_measures():
    e = ENV["envelope"]
    assert ENV["trials"] == 100 and ENV["steps_per_trial"] == 300 and len(ENV["orders"]) == 5
    # the solver order cannot move a kinematic body: exact for every order, over 802 x 4 episodes
    assert e["ship_pose_max_abs"] == 0 and e["termination_step_mismatches"] == 0 and e["success_mismatches"] == 0
    # and what it can move stays small on average while single trajectories diverge (contact dynamics are chaotic)
    assert e["total_work_episode_mean_rel"] < 0.01 and e["batch_mean_total_work_rel"] < 0.01
    assert 0.0 < e["floe_position_max_abs_m_after_k_steps"]["1"] < e["floe_position_max_abs_m_after_k_steps"]["20"]


def test_50pct_envelope_has_the_same_exact_quantities():
    e50 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "order_envelope_50pct.json")))["envelope"]
    assert e50["ship_pose_max_abs"] == 0 and e50["termination_step_mismatches"] == 0 and e50["success_mismatches"] == 0
    assert e50["total_work_episode_mean_rel"] < 0.01 and e50["batch_mean_total_work_rel"] < 0.01


def test_sample_stays_inside_the_envelope():
    trials = default_trials(0.3, 6, base_seed=0)
    jobs = [(trials[i], i, 24, mode) for mode in (0, 1, 3) for i in range(6)]
    with ThreadPoolExecutor(6) as ex:
        res = list(ex.map(_run, jobs))
    base = res[:6]
    e = ENV["envelope"]
    grow = e["floe_position_max_abs_m_after_k_steps"]
    for m, alt in ((1, res[6:12]), (3, res[12:18])):
        for (r0, s0), (r1, s1) in zip(base, alt):
            assert r0.shape == r1.shape                                    # same termination step
            assert np.array_equal(r0[:, :5], r1[:, :5])                    # ship pose, terminated, success: bit-identical
            assert np.abs(r0[:, 6] - r1[:, 6]).max() <= e["step_reward_max_abs"]
            for k in s0:
                if k in s1 and str(k) in grow:
                    assert np.abs(s0[k] - s1[k]).max() <= grow[str(k)] * 1.0 + 1e-12, (m, k)
