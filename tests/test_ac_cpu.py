"""area-clearing-v0 oracle: layouts, shapely-predicate restatements against brute force, goal map rules, env-level known answers."""
import math

import numpy as np
import pytest

from benchpush_amd import area_clearing_scenario as A
from benchpush_amd.config import default_cfg
from benchpush_amd.metrics.task_driven_metric import TaskDrivenMetric, _mst_weight
from oracle.oracle_bd import OracleAreaClearing


def _oracle(cfg, trial):
    o = OracleAreaClearing(A.area_clearing_physics_params(cfg), A.area_clearing_params(cfg), cfg)
    o.reset(trial, observe=False)
    return o


def test_layout_follows_the_reference_draw_order():
    import random
    cfg = default_cfg("area_clearing")
    t = A.generate_trial(cfg, 5)
    rng = random.Random(5)
    assert t["start"][0] == (-5 + 1) + rng.random() * (10 - 2) and t["start"][1] == -4.0
    assert len(t["boxes"]) == 10 and len(t["statics"][1]) == 4
    d = np.linalg.norm(t["boxes"][:, None, :2] - t["boxes"][None, :, :2], axis=2) + np.eye(10) * 9
    assert d.min() > cfg.min_obs_dist
    g = A.goal_points(cfg)
    assert g.shape == (40, 2) and np.allclose(g[0], [-5.0, -4.5]) and np.allclose(g[10], [-4.5, 5.0])
    cfg.env = "walled_env_with_columns"
    assert len(A.static_shapes(cfg)[1]) == 2 + 4 + 3


def test_goal_map_and_cspace_rules():
    cfg = default_cfg("area_clearing")
    o = _oracle(cfg, A.generate_trial(cfg, 0))
    m = o.maps()
    H, W = o.H, o.W
    assert (H, W) == (468, 468)
    ppm = o.bd["ppm"]
    # padded room is free outside the small map, walls are dilated by floor(int(2 r ppm) / 4) = 7 px
    assert m["cspace"][0, 0] == 1 and m["cspace"][H // 2, W // 2] == 1
    i8 = int(np.floor(H / 2 - 8.0 * ppm))                  # row of the outer wall face y = +8 m
    assert m["cspace"][i8 + 9, W // 2] == 1 and m["cspace"][i8 + 6, W // 2] == 0   # wall rows <= i8 + 1, dilated by 7
    g = m["recept"]
    assert g[H // 2, W // 2] > 0 and g[H // 2, W // 2] <= 0.5   # inside the clearance boundary: scaled distance in (0, 0.5]
    j6 = int(np.floor(W / 2 + 6.5 * ppm))
    assert g[H // 2, j6] == 0.0                            # between boundary and outer boundary: 0
    assert g[0, 0] == 1.0 and g[i8 - 5, W // 2] == 2.0     # outside the outer boundary: 1 (+1 where the cell is blocked)
    assert g[i8 + 3, W // 2] == 1.0                        # blocked cell between the boundaries: 0 + 1


def test_observation_classes_and_clearing_a_box():
    cfg = default_cfg("area_clearing")
    tr = dict(A.generate_trial(cfg, 0))
    tr["start"] = np.array([0.0, 2.0, np.pi / 2])
    tr["boxes"] = np.array([[0.0, 4.2, 0.0], [-3.0, -3.0, 0.0]])
    cfg.num_obstacles = 2
    o = _oracle(cfg, tr)
    obs = o.observe()
    assert set(np.unique(obs[..., 0])).issubset({0, 31, 95, 127, 159, 223}) and (obs[..., 0] == 159).sum() > 0
    total = 0.0
    cleared_at = None
    saw_completed = False
    for t in range(6):
        obs, r, term, trunc, info = o.step(1.0)            # straight ahead: pushes box 0 over the boundary at y = 5
        total += r
        saw_completed = saw_completed or bool((obs[..., 0] == 223).any())
        if info["box_count"] == 1 and cleared_at is None:
            cleared_at = t
        assert np.isfinite(r) and info["t"] == (t + 1 if cleared_at is None else t - cleared_at)
        if cleared_at == t:
            assert r > 9 and info["t"] == 0                 # BOX_CLEARED_REWARD (+ pushing term), time counter reset
    assert cleared_at is not None and not term
    assert saw_completed                                   # a completed cube is drawn in its own class (7/8) until the robot covers it


def test_task_driven_metric_against_networkx():
    import networkx as nx
    rng = np.random.RandomState(0)
    for _ in range(10):
        n = rng.randint(2, 8)
        edges = [(i, j, float(rng.rand())) for i in range(n) for j in range(i + 1, n) if rng.rand() < 0.7]
        G = nx.Graph(); G.add_nodes_from(range(n))
        for a, b, w in edges:
            G.add_edge(a, b, weight=w)
        ref = sum(d["weight"] for _, _, d in nx.minimum_spanning_tree(G).edges(data=True))
        assert abs(_mst_weight(n, edges) - ref) < 1e-12
    m = TaskDrivenMetric("x", 1.0)
    boxes = [np.array([[6, 6], [7, 6], [7, 7], [6, 7.0]]), np.array([[0, 0], [1, 0], [1, 1], [0, 1.0]])]
    m.reset({"state": (0, 0, 0), "obs": boxes, "goal_positions": [(5, 6.5), (5, 0)]})
    m.update({"total_work": 1.0, "box_completed_statuses": [True, False], "state": (3, 4, 0)}, 1.0, eps_complete=True)
    assert abs(m.efficiency_scores[0] - (np.hypot(6.5, 6.5) + 1.5) / 5.0) < 1e-12 and m.success_rates == [0.5]
    assert abs(m.effort_scores[0] - (5.0 + 1.5 * 1.0) / (5.0 + 1.0)) < 1e-12


def _wall_cut_cfg():
    """walled_env plus a wall that crosses the left edge of the clearance boundary (x = -5) at y = 0: area_clearing.py:236-240 then removes the
    wall's 0.1 m buffer from that edge (`LineString.difference`) and the boundary has five goal lines instead of four."""
    cfg = default_cfg("area_clearing")
    cfg.env = "walled_env"
    cfg.envs.walled_env.walls = list(cfg.envs.walled_env.walls) + [[[-7.0, 0.0], [-3.0, 0.0]]]
    return cfg


def test_walls_that_cut_the_clearance_boundary():
    """_compute_boundary_goals with a wall across an edge: the edge is split at the wall's buffer, pieces keep the edge's
    direction and place, 10 goal points per piece; an oblique wall cuts at the exact intersection with its offset lines; a wall whose rounded
    end reaches an edge cuts it where the edge crosses GEOS' 16-chords-per-quadrant polygon of the cap; pieces of 0.1 m or less are dropped."""
    cfg = _wall_cut_cfg()
    lines = A.boundary_goal_lines(cfg)
    assert lines[0] == ([-5.0, -5.0], [-5.0, -0.1]) and lines[1] == ([-5.0, 0.1], [-5.0, 5.0]) and len(lines) == 5
    g = A.goal_points(cfg)
    assert g.shape == (50, 2) and np.allclose(g[0], [-5.0, -5.0 + 0.05 * 4.9]) and np.allclose(g[10], [-5.0, 0.1 + 0.05 * 4.9]) and np.allclose(g[20], [-4.5, 5.0])
    # oblique wall through the top edge y = 5: the buffer's straight sides are the lines through (x, 5) at distance 0.1 from the wall's axis
    a, b = (0.0, 3.0), (2.0, 7.0)
    (l0, l1), (r0, r1) = A.cut_edge_by_wall([-5.0, 5.0], [5.0, 5.0], (a, b))
    for q in (l1, r0):
        assert q[1] == 5.0 and abs(A._dist_to_segment(q[0], q[1], a, b) - 0.1) < 1e-12
    assert l0 == [-5.0, 5.0] and r1 == [5.0, 5.0] and l1[0] < 1.0 < r0[0]
    # wall that does not reach the edge: untouched; wall along the edge: the covered part disappears (one end inside the buffer)
    assert A.cut_edge_by_wall([-5.0, 5.0], [5.0, 5.0], ((0.0, 0.0), (0.0, 4.0))) == [([-5.0, 5.0], [5.0, 5.0])]
    # the cap, not a straight side, reaches the edge: the cut is where the edge crosses GEOS' polygon of the cap (16 chords per quadrant),
    # at most r (1 - cos(pi / 64)) = 0.12 mm inside the circle of radius 0.1 around the wall's end, and on the edge exactly
    wall = ((0.0, 0.0), (0.0, 4.95))
    (l0, l1), (r0, r1) = A.cut_edge_by_wall([-5.0, 5.0], [5.0, 5.0], wall)
    half = math.sqrt(0.1 ** 2 - 0.05 ** 2)
    assert l0 == [-5.0, 5.0] and r1 == [5.0, 5.0] and l1[1] == 5.0 and r0[1] == 5.0
    for q, want in ((l1, -half), (r0, half)):
        assert 0.0 <= abs(want) - abs(q[0]) <= 2.5e-4        # along the edge; the radial bound is the next line
        assert 0.1 - 1.3e-4 <= A._dist_to_segment(q[0], q[1], *wall) <= 0.1
    # the polygon itself: offset point, 31 fan points and offset point per end, all on the circle / the offset lines
    ring = A.segment_buffer_ring(wall)
    assert len(ring) == 66 and ring[0] == (-0.1, 4.95) and ring[32] == (0.1, 4.95) and ring[33] == (0.1, 0.0) and ring[65] == (-0.1, 0.0)
    assert all(abs(A._dist_to_segment(x, y, *wall) - 0.1) < 1e-15 for x, y in ring)
    assert abs(ring[16][0]) < 1e-16 and abs(ring[16][1] - 5.05) < 1e-15                      # the fan's middle point: straight beyond the wall's end
    # an oblique wall whose end lies 0.07 below the edge: both cuts on the cap polygon, symmetric about the foot of the end point only in the circle limit
    (ol0, ol1), (or0, or1) = A.cut_edge_by_wall([-5.0, 5.0], [5.0, 5.0], ((1.0, 3.0), (2.0, 4.93)))
    assert ol1[1] == 5.0 and or0[1] == 5.0 and ol1[0] < 2.0 < or0[0]
    for q in (ol1, or0):
        assert 0.1 - 1.3e-4 <= math.hypot(q[0] - 2.0, q[1] - 4.93) <= 0.1 + 1e-15
    cfg3 = default_cfg("area_clearing")
    cfg3.env = "walled_env"
    cfg3.envs.walled_env.walls = list(cfg3.envs.walled_env.walls) + [[[-4.95, 0.0], [-3.0, 0.0]]]
    lines3 = A.boundary_goal_lines(cfg3)
    assert len(lines3) == 5 and lines3[0][0] == [-5.0, -5.0] and lines3[0][1][0] == -5.0 and lines3[1][0][0] == -5.0 and lines3[1][1] == [-5.0, 5.0]
    assert abs(lines3[0][1][1] + half) < 1.3e-4 and abs(lines3[1][0][1] - half) < 1.3e-4 and A.goal_points(cfg3).shape == (50, 2)
    # two walls close together leave a piece of 0.05 m between their buffers: dropped (area_clearing.py:246-250)
    cfg2 = default_cfg("area_clearing")
    cfg2.env = "walled_env"
    cfg2.envs.walled_env.walls = list(cfg2.envs.walled_env.walls) + [[[-7.0, 0.0], [-3.0, 0.0]], [[-7.0, 0.25], [-3.0, 0.25]]]
    lines2 = A.boundary_goal_lines(cfg2)
    assert lines2[0] == ([-5.0, -5.0], [-5.0, -0.1]) and lines2[1] == ([-5.0, 0.35], [-5.0, 5.0]) and len(lines2) == 5


def test_wall_cut_layout_runs_through_the_oracle():
    """The oracle accepts the 50 goal points of the wall-cut layout and its goal-distance map sees them (channel 3 of the observation)."""
    cfg = _wall_cut_cfg()
    o = _oracle(cfg, A.generate_trial(cfg, 1))
    m = o.maps()
    assert np.isfinite(m["recept"]).all() and m["recept"].max() > 0
    obs, r, term, trunc, info = o.step(0.3)
    assert obs.shape == (224, 224, 4) and np.isfinite(r)
