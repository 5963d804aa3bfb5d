"""Episode generation, local-map extraction and path execution pinned against golden vectors produced by the reference's own
BoxDeliveryEnv / AreaClearingEnv classes (tests/golden/make_golden_env_methods.py)."""
import json
import os

import numpy as np
import pytest

from benchpush_amd import area_clearing_scenario as A
from benchpush_amd import box_delivery_scenario as S
from benchpush_amd.config import default_cfg
from oracle.oracle_bd import OracleAreaClearing, OracleBoxDelivery

HERE = os.path.dirname(__file__)
G = json.load(open(os.path.join(HERE, "golden", "env_methods_golden.json")))
Z = np.load(os.path.join(HERE, "golden", "env_methods_golden.npz"))


@pytest.mark.parametrize("oc", ["small_empty", "small_columns", "large_columns", "large_divider"])
def test_episode_generator_matches_reference_random_stream(oc):
    cfg = default_cfg("box_delivery")
    cfg.env.obstacle_config = oc
    trials = S.generate_trials(cfg, 3)
    for t, ref in zip(trials, G["box_delivery_episodes"][oc]):
        assert list(t["start"]) == ref["start"]
        assert t["boxes"].tolist() == ref["boxes"]
        assert len(t["boundary"]) == len(ref["boundary"])
        for b, r in zip(t["boundary"], ref["boundary"]):
            assert b["type"] == r["type"] and [float(v) for v in b["position"]] == r["position"]
            if r["vertices"] is not None:
                assert np.asarray(b["vertices"], np.float64).tolist() == r["vertices"]
            if r["heading"] is not None:
                assert float(b["heading"]) == r["heading"]


def _bd_oracle(start, boxes):
    cfg = default_cfg("box_delivery")
    tr = dict(S.generate_trials(cfg, 1)[0])
    tr["start"] = np.array(start, np.float64)
    tr["boxes"] = np.array(boxes, np.float64)
    bp = S.box_delivery_params(cfg)
    bp["num_boxes"] = len(boxes)
    o = OracleBoxDelivery(S.box_delivery_physics_params(cfg), bp, cfg)
    o.reset(tr, observe=False)
    return cfg, o


def test_robot_state_channel_room_shape_and_local_maps():
    cfg, o = _bd_oracle([0.0, 0.0, 0.0], [[4.5, -2.2, 0.0]])
    assert [o.H, o.W] == G["padded_room_shape"] and abs(o.bd["robot_radius"] - G["robot_radius"]) < 1e-15
    rsc = np.unpackbits(np.array(G["robot_state_channel_rows"], np.uint8), axis=1)[:, :224]
    assert np.array_equal(o.observe()[..., 1], rsc * 255)
    gm = (np.random.RandomState(5).randint(0, 9, (o.H, o.W)) / 8).astype(np.float32)
    for k, (x, y, h) in enumerate(G["local_map_poses"]):
        lm = o.local_map(gm, x, y, h)
        assert int(((lm * 8).astype(np.uint8) != Z["local%d_map_u8" % k]).sum()) <= 2      # cosdg vs deterministic sincos on the border
        ld = lm - lm.min()
        assert int(((ld * 8).astype(np.uint8) != Z["local%d_dist_u8" % k]).sum()) <= 2


def test_execute_robot_path_matches_reference_loop_in_free_space():
    """The reference's loop ran on a stand-in space that integrates the kinematic robot alone; here the full restated physics runs
    with the boxes parked far from the path.  libm vs the deterministic atan2/sincos differ by ulps, hence the tolerances."""
    checked = 0
    for c in G["execute_robot_path"]:
        pts = np.array([w[:2] for w in c["waypoints"]])
        if c["task"] == "box_delivery":
            if np.abs(pts[:, 0]).max() > 3.8 or np.abs(pts[:, 1]).max() > 1.6:
                continue                                      # the stand-in space has no walls: keep to paths inside the room
            _, o = _bd_oracle(c["start"], [[4.6, -2.2, 0.0], [-4.6, -2.2, 0.0]])
        else:
            cfg = default_cfg("area_clearing")
            cfg.num_obstacles = 2
            tr = dict(A.generate_trial(cfg, 0))
            tr["start"] = np.array(c["start"], np.float64)
            tr["boxes"] = np.array([[6.5, 6.5, 0.0], [-6.5, 6.5, 0.0]])
            o = OracleAreaClearing(A.area_clearing_physics_params(cfg), A.area_clearing_params(cfg), cfg)
            o.reset(tr, observe=False)
        r = o.execute_path(c["waypoints"])
        assert r["sim_steps"] == c["sim_steps"], (c["task"], r["sim_steps"], c["sim_steps"])
        assert abs(r["robot_distance"] - c["robot_distance"]) < 1e-9
        assert np.abs(r["final"] - np.array(c["final"])).max() < 1e-9
        assert abs(r["turn_angle"] - c["turn_angle"]) < 1e-9
        checked += 1
    assert checked >= 5
