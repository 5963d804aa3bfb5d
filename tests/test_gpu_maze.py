"""maze-NAMO-v0 parity tests (-m gpu): HIP path through the C ABI against the CPU oracle, bit for bit."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _oracles(env, n):
    from oracle.oracle import OracleMaze
    c = env.cfg
    return [OracleMaze(env.params, c.robot.vertices, c.robot.wheel_vertices, c.obstacle_size) for _ in range(n)]


def _run(E, nbox, T, steps, seed, action_fn=None, **cfgkw):
    from benchpush_amd.envs.maze_namo import BatchedMazeEnv
    env = BatchedMazeEnv(E, cfg=dict({"num_obstacles": nbox}, **cfgkw), num_layouts=T, base_seed=seed, device="cuda:0")
    obs, info = env.reset()
    orcs = _oracles(env, E)
    eps = [0] * E
    for e, o in enumerate(orcs):
        assert np.array_equal(obs[e].cpu().numpy(), o.reset(env.layouts[e % T])), ("reset obs", e)
    rng = np.random.default_rng(seed)
    nterm = 0
    for t in range(steps):
        a = rng.uniform(-1, 1, E) if action_fn is None else np.array([action_fn(e, t) for e in range(E)], np.float64)
        obs, rew, term, trunc, info = env.step(torch.from_numpy(a))
        bs = env.body_state().cpu().numpy()
        go, gi, gr, gt = obs.cpu().numpy(), info.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
        for e, o in enumerate(orcs):
            oo, orr, ot, oi = o.step(float(a[e]))
            ss = o.shape_states()
            assert np.array_equal(bs[e, : len(ss)], ss), ("state", t, e)
            assert np.array_equal(go[e], oo), ("obs", t, e)
            assert np.array_equal(gi[e], np.array(list(oi.values()))), ("info", t, e)
            assert gr[e] == orr and bool(gt[e]) == ot, ("reward/term", t, e)
        done = gt.astype(bool)
        if done.any():
            nterm += int(done.sum())
            obs, info = env.reset(term)
            for e in np.nonzero(done)[0]:
                eps[e] += 1
                assert np.array_equal(obs[e].cpu().numpy(), orcs[e].reset(env.layouts[(e + eps[e]) % T])), ("auto-reset obs", t, e)
    env.check_errors()
    return nterm


def test_parity_20_boxes_random_actions_with_resets():
    rng = np.random.default_rng(11)
    acts = rng.uniform(-1, 1, (70, 8))
    acts[:, 0], acts[:, 1] = 1.0, -1.0      # hard-over: these two end on a wall (-50, no success) and are auto-reset
    assert _run(E=8, nbox=20, T=3, steps=70, seed=0, action_fn=lambda e, t: float(acts[t, e])) >= 2


def test_parity_default_5_boxes_and_maze_v2():
    _run(E=3, nbox=5, T=2, steps=25, seed=4)
    _run(E=2, nbox=8, T=2, steps=15, seed=9, maze_version=2)


def test_parity_random_start_per_layout():
    """cfg.random_start (maze_NAMO_env.py:229-238): every layout carries its own rejection-sampled start (heading 3 pi / 2), drawn before
    the boxes from the layout's generator; the envs start there and match the oracle through resets."""
    from benchpush_amd.envs.maze_namo import BatchedMazeEnv
    from benchpush_amd.maze_scenario import point_query_hits_wall
    env = BatchedMazeEnv(2, cfg={"num_obstacles": 6, "random_start": True, "maze_version": 2}, num_layouts=3, base_seed=3, device="cuda:0")
    starts = [tuple(l["start"]) for l in env.layouts]
    assert len(set(starts)) == 3 and all(abs(s[2] - 1.5 * math.pi) < 1e-15 and 1 <= s[0] <= 20 and 1 <= s[1] <= 20 for s in starts)
    assert not any(point_query_hits_wall(env.walls, s[0], s[1], env.cfg.robot.min_obstacle_dist) for s in starts)
    _, info = env.reset()
    assert np.allclose(info[:, :3].cpu().numpy(), np.array(starts[:2]), atol=1e-9)   # the settle leaves the kinematic robot in place
    env.close()
    _run(E=3, nbox=6, T=3, steps=12, seed=3, random_start=True, maze_version=2)


def test_maze_damping_and_large_hulls_do_not_get_the_ship_yaw_clamp():
    """A maze handle served by a generic KIND 0 kernel -- `sim.damping != 0` (k_physics_step_damp) or a robot outline above 8 vertices (k_physics_step) --
    keeps MazeNAMO.step's rules (maze_NAMO_env.py:405-419: boundary only): the ship's yaw clamp (ship_ice_env.py:284-287) must not zero the commanded
    angular velocity when the heading is outside (0, pi).  maze_version 2 starts at heading 3 pi / 2, i.e. outside from the first sub-step; the v1 envs turn
    hard over until they pass pi / 0."""
    _run(E=3, nbox=6, T=2, steps=12, seed=5, maze_version=2, sim={"damping": 0.8})
    _run(E=2, nbox=5, T=2, steps=26, seed=6, action_fn=lambda e, t: 1.0 if e == 0 else -1.0, sim={"damping": 0.5})
    # a 10-vertex outline (the shipped octagon with two extra vertices on its long sides): hulls above 8 vertices take the generic instantiation
    from benchpush_amd.config import default_cfg
    verts = [list(map(float, v)) for v in default_cfg("maze_namo").robot.vertices]
    hull = np.array(verts)
    i0 = int(np.argmax(np.linalg.norm(np.roll(hull, -1, 0) - hull, axis=1)))           # longest edge and the one opposite: bulge their midpoints outwards
    ext = []
    for k, v in enumerate(verts):
        ext.append(v)
        if k in (i0, (i0 + len(verts) // 2) % len(verts)):
            a, b = hull[k], hull[(k + 1) % len(verts)]
            mid, e = (a + b) / 2, b - a
            nrm = np.array([e[1], -e[0]]) / np.linalg.norm(e)
            if np.dot(nrm, mid - hull.mean(0)) < 0:
                nrm = -nrm
            ext.append(list(mid + 0.01 * nrm))
    assert len(ext) == len(verts) + 2
    _run(E=2, nbox=6, T=2, steps=10, seed=7, maze_version=2, robot={"vertices": ext})


def test_parity_straight_drive_pushes_boxes():
    _run(E=2, nbox=20, T=2, steps=30, seed=2, action_fn=lambda e, t: 0.0)


def test_gym_adapter_and_metric():
    import benchpush_amd
    from benchpush_amd.metrics import MazeNamoMetric
    env = benchpush_amd.make("maze-NAMO-v0", cfg={"num_obstacles": 6}, num_layouts=2)
    u = env.unwrapped
    assert u.observation_space.shape == (4, 192, 192) and u.goal == (3.75, 3.75)
    assert abs(u.max_yaw_rate_step - (math.pi / 2) / 15) < 1e-15
    obs, info = u.reset()
    assert obs.shape == (4, 192, 192) and obs.dtype == np.uint8
    assert set(info) == {"state", "total_work", "obs", "box_count", "goal_dt", "m_to_pix_scale"} and len(info["obs"]) == 6
    assert info["goal_dt"].shape == (240, 240) and info["m_to_pix_scale"] == 16
    metric = MazeNamoMetric("test", robot_mass=u.cfg.robot.mass)
    metric.reset(info)
    assert metric.L > 10   # metres along the wavefront from the start to the goal
    for t in range(200):
        obs, r, done, trunc, info = u.step(np.float64(0.4))
        assert set(info) == {"state", "total_work", "collision reward", "scaled collision reward", "dist increment reward",
                             "trial_success", "obs"}
        metric.update(info, r, done or trunc)
        if done:
            break
    assert done and len(metric.effort_scores) == 1
    env.close()


def test_full_size_properties_4096_envs():
    """BASELINE.json configs[2] size (4096 envs, 20 boxes): oracle-free properties."""
    from benchpush_amd.envs.maze_namo import BatchedMazeEnv
    E, T = 4096, 8
    env = BatchedMazeEnv(E, cfg={"num_obstacles": 20}, num_layouts=T, base_seed=1, device="cuda:0")
    obs, info = env.reset()
    g = torch.Generator(device="cuda:0")
    g.manual_seed(3)
    base = torch.rand((8, T), generator=g, device="cuda:0", dtype=torch.float64) * 2 - 1
    prev = torch.zeros(E, dtype=torch.float64, device="cuda:0")
    for t in range(8):
        obs, rew, term, trunc, info = env.step(base[t].repeat(E // T))
        v = obs.view(E // T, T, -1)
        assert torch.equal(v, v[0:1].expand(E // T, -1, -1))                       # same layout + actions -> same bits
        assert (info[:, 4] >= 0).all() and (info[:, 3] >= prev).all() and torch.isfinite(info).all()
        prev = info[:, 3].clone()
        assert ((obs[:, 2] == 0) | (obs[:, 2] <= 255)).all()
        assert (obs[:, 0].view(E, -1).max(dim=1).values == 255).all()              # the robot is always in its own view
        env.reset(term)
    env.check_errors()


def test_scheduler_parks_and_resumes_maze_envs_bit_identically(monkeypatch):
    """k_physics_step_sched_maze at full size (4096 envs x 60 steps with auto-reset, thousands of envs parked and resumed): the default scheduler and a
    ragged chunk size leave every env in exactly the state of the one-env-per-wavefront kernel (BP_SCHED=0) -- body state (incl. the five kinematic robot
    slots reloaded from velv), rewards, termination (sticky wall flag carried through D.sq_carry), observations, info and episode metrics."""
    from benchpush_amd.envs.maze_namo import BatchedMazeEnv
    E, T, steps = 4096, 16, 60
    g = torch.Generator(device="cuda:0")
    g.manual_seed(9)
    acts = (torch.rand((steps, E), generator=g, device="cuda:0", dtype=torch.float64) * 2 - 1).float().double()
    acts[:, 0::4] = 1.0                                        # hard-over: these envs end on a wall and restart inside the window
    acts[:, 1::4] = -1.0

    def run(env_vars):
        monkeypatch.delenv("BP_SCHED", raising=False)
        for k, v in env_vars.items():
            monkeypatch.setenv(k, v)
        env = BatchedMazeEnv(E, cfg={"num_obstacles": 20}, num_layouts=T, base_seed=2, device="cuda:0")
        env.reset()
        rsum = torch.zeros(E, dtype=torch.float64, device="cuda:0")
        nterm = 0
        for t in range(steps):
            obs, rew, term, _, info = env.step(acts[t])
            rsum += rew
            nterm += int(term.sum().item())
            env.reset(term)
        env.check_errors()
        out = (env.body_state().clone(), rsum, env.obs.clone(), env.info.clone(), env.episode_metrics()[0].clone(), nterm)
        env.close()
        return out

    ref = run({"BP_SCHED": "0"})
    assert ref[5] > 100                                        # episodes ended (wall hits) and restarted inside the window
    for variant in ({}, {"BP_SCHED": "37"}):
        got = run(variant)
        for a, b in zip(ref[:5], got[:5]):
            assert torch.equal(a, b), variant
        assert got[5] == ref[5]


def test_episode_metrics_equal_host_maze_namo_metric():
    """On-device episode rows (bp_get_episode_metrics) == MazeNamoMetric.reset / update (maze_namo_metric.py:25-75) fed step by step:
    L from the wavefront map at the rounded start pixel, path length from the rounded state, effort with the robot's mass."""
    from benchpush_amd.envs.maze_namo import BatchedMazeEnv
    from benchpush_amd.metrics import MazeNamoMetric
    E = 4
    env = BatchedMazeEnv(E, cfg={"num_obstacles": 8}, num_layouts=3, base_seed=1, device="cuda:0")
    gdt, s = env.goal_map(), float(env.cfg.occ.m_to_pix_scale)
    host = [MazeNamoMetric("x", robot_mass=float(env.cfg.robot.mass)) for _ in range(E)]

    def info_dict(row, reset=False):
        d = {"state": (round(float(row[0]), 2), round(float(row[1]), 2), round(float(row[2]), 2)), "total_work": float(row[3]),
             "trial_success": bool(row[8])}
        if reset:
            d.update(goal_dt=gdt, m_to_pix_scale=s)
        return d

    _, info = env.reset()
    for e in range(E):
        host[e].reset(info_dict(info[e].cpu().numpy(), reset=True))
    rng = np.random.default_rng(2)
    lengths, finished, nrows = np.zeros(E, int), np.zeros(E, int), 0
    for t in range(60):
        a = rng.uniform(-1, 1, E)
        a[0], a[1] = 1.0, -1.0                                  # hard-over: these end on a wall and restart
        _, rew, term, _, info = env.step(torch.from_numpy(a))
        inf, rw, tm = info.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy().astype(bool)
        rows, cnt = env.episode_metrics()
        rows, cnt = rows.cpu().numpy(), cnt.cpu().numpy()
        lengths += 1
        for e in range(E):
            host[e].update(info_dict(inf[e]), float(rw[e]), eps_complete=bool(tm[e]))
            if tm[e]:
                finished[e] += 1
                nrows += 1
                assert cnt[e] == finished[e] and rows[e, 2] == host[e].rewards[-1] and rows[e, 3] == float(inf[e, 8])
                assert rows[e, 4] == lengths[e] and rows[e, 5] == inf[e, 3]
                assert math.isclose(rows[e, 0], host[e].efficiency_scores[-1], rel_tol=1e-12, abs_tol=0.0)
                assert math.isclose(rows[e, 1], host[e].effort_scores[-1], rel_tol=1e-12, abs_tol=0.0)
        if tm.any():
            _, info2 = env.reset(term)
            for e in np.nonzero(tm)[0]:
                host[e].reset(info_dict(info2[e].cpu().numpy(), reset=True))
                lengths[e] = 0
    assert nrows >= 2
    env.check_errors()
    env.close()
