"""-m gpu: on-device episode metrics (bp_get_episode_metrics) against the host ShipIceMetric fed step by step, the device's
round(x, 2), per-episode random starts against the oracle, and the whole-episode soak (capacity flags deep in episodes)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(E, conc, trials, **kw):
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv
    return BatchedShipIceEnv(E, cfg=dict({"concentration": conc}, **kw), trials=trials, device="cuda:0")


def test_device_round2_equals_python_round():
    """info['state'] is rounded with python's round(x, 2) (ship_ice_env.py:337-339); the metric path length is integrated from it."""
    from benchpush_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-3, 45, 200000), np.arange(-200, 4200) / 100.0, (np.arange(-2000, 42000) + 0.5) / 1000.0,
                         np.array([0.125, 0.375, 2.675, 1.005, 0.285, -0.125, 0.0, -0.0, 1e-9, 12.345, 39.995, 8.994999999999999])])
    t = torch.from_numpy(xs).to("cuda:0")
    out = torch.empty_like(t)
    assert L.bp_debug_round2(t.data_ptr(), out.data_ptr(), t.numel(), None) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    want = np.array([round(float(v), 2) for v in xs])
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:5]


def test_episode_metrics_equal_host_ship_ice_metric():
    """Device accumulators == ShipIceMetric.reset / update (ship_ice_metric.py:26-69) driven with the same info dicts; rows appear when
    an env terminates, and a reset of a running episode closes it as truncated (eps_complete = truncated)."""
    from benchpush_amd.envs.ship_ice import default_trials
    from benchpush_amd.metrics import ShipIceMetric
    E, T = 6, 3
    trials = default_trials(0.3, T, base_seed=11, goal_y=2.6)
    env = _mk(E, 0.3, trials, goal_y=2.6)
    mass = float(env.cfg.ship.mass)
    host = [ShipIceMetric("x", ship_mass=mass, goal=env.goal) for _ in range(E)]

    def info_dict(row):
        return {"state": (round(float(row[0]), 2), round(float(row[1]), 2), round(float(row[2]), 2)), "total_work": float(row[3]),
                "trial_success": bool(row[8])}

    _, info = env.reset()
    inf = info.cpu().numpy()
    for e in range(E):
        host[e].reset(info_dict(inf[e]))
    rng = np.random.default_rng(3)
    finished = np.zeros(E, int)
    lengths = np.zeros(E, int)
    nrows = 0
    for t in range(40):
        a = rng.uniform(-0.4, 0.4, E)
        _, rew, term, _, info = env.step(torch.from_numpy(a))
        inf, rw, tm = info.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy().astype(bool)
        rows, cnt = env.episode_metrics()
        rows, cnt = rows.cpu().numpy(), cnt.cpu().numpy()
        lengths += 1
        for e in range(E):
            host[e].update(info_dict(inf[e]), float(rw[e]), eps_complete=bool(tm[e]))
            if tm[e]:
                finished[e] += 1
                assert cnt[e] == finished[e]
                assert rows[e, 2] == host[e].rewards[-1] and rows[e, 3] == float(inf[e, 8])
                assert rows[e, 4] == lengths[e] and rows[e, 5] == inf[e, 3]
                assert math.isclose(rows[e, 0], host[e].efficiency_scores[-1], rel_tol=1e-12, abs_tol=0.0)
                assert math.isclose(rows[e, 1], host[e].effort_scores[-1], rel_tol=1e-12, abs_tol=0.0)
                nrows += 1
            else:
                assert cnt[e] == finished[e]
        mask = torch.from_numpy(tm.astype(np.uint8))
        if t == 20:  # truncate env 0 mid-episode: the reset closes its episode with success 0
            if not tm[0]:
                host[0].update(info_dict(inf[0]), 0.0, eps_complete=False)  # no-op on the scores; the lists below are made by hand
                eff, effort, r = 0.0, host[0].compute_effort_score(), host[0].eps_reward
                mask[0] = 1
                _, info2 = env.reset(mask)
                rows2, cnt2 = env.episode_metrics()
                rows2, cnt2 = rows2.cpu().numpy(), cnt2.cpu().numpy()
                finished[0] += 1
                assert cnt2[0] == finished[0] and rows2[0, 0] == eff and rows2[0, 3] == 0.0 and rows2[0, 2] == r
                assert math.isclose(rows2[0, 1], effort, rel_tol=1e-12) and rows2[0, 4] == lengths[0]
                inf2 = info2.cpu().numpy()
                for e in range(E):
                    if mask[e]:
                        host[e].reset(info_dict(inf2[e]))
                        lengths[e] = 0
                continue
        if tm.any():
            _, info2 = env.reset(mask)
            inf2 = info2.cpu().numpy()
            for e in range(E):
                if tm[e]:
                    host[e].reset(info_dict(inf2[e]))
                    lengths[e] = 0
    assert nrows >= 6
    env.check_errors()
    env.close()


def test_episode_lists_equal_host_metric_lists_without_per_step_sync():
    """bp_get_episode_history: the ring of the last episodes and the running sums equal the LISTS ShipIceMetric keeps (base_metric.py:12-16,
    ship_ice_metric.py:57-60) over >= 3 episodes per env, with the device never consulted between steps: the loop only records what it feeds the host
    metric; the device buffers are read once at the end."""
    from benchpush_amd.envs.ship_ice import default_trials
    from benchpush_amd.metrics import ShipIceMetric
    E, T = 5, 3
    trials = default_trials(0.3, T, base_seed=21, goal_y=2.2)
    env = _mk(E, 0.3, trials, goal_y=2.2)
    mass = float(env.cfg.ship.mass)
    host = [ShipIceMetric("x", ship_mass=mass, goal=env.goal) for _ in range(E)]

    def info_dict(row):
        return {"state": (round(float(row[0]), 2), round(float(row[1]), 2), round(float(row[2]), 2)), "total_work": float(row[3]),
                "trial_success": bool(row[8])}

    _, info = env.reset()
    for e in range(E):
        host[e].reset(info_dict(info[e].cpu().numpy()))
    rng = np.random.default_rng(4)
    lengths, all_len = np.zeros(E, int), [[] for _ in range(E)]
    for t in range(90):
        a = rng.uniform(-0.3, 0.3, E)
        _, rew, term, _, info = env.step(torch.from_numpy(a))
        inf, rw, tm = info.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy().astype(bool)   # (the reset below rewrites the info rows it resets)
        _, info2 = env.reset(term)                         # device mask; rows of finished envs restart
        inf2 = info2.cpu().numpy()
        lengths += 1
        for e in range(E):
            host[e].update(info_dict(inf[e]), float(rw[e]), eps_complete=bool(tm[e]))
            if tm[e]:
                all_len[e].append(int(lengths[e])); lengths[e] = 0
                host[e].reset(info_dict(inf2[e]))
    ring, sums, cnt = env.episode_history()
    lists = env.episode_lists()
    ring, sums, cnt = ring.cpu().numpy(), sums.cpu().numpy(), cnt.cpu().numpy()
    assert cnt.min() >= 3 and cnt.max() > 8                     # enough episodes, and the ring has wrapped for some env
    for e in range(E):
        n = int(cnt[e])
        assert n == len(host[e].rewards) == len(host[e].efficiency_scores) == len(host[e].effort_scores) == len(all_len[e])
        got = lists[e]
        k0 = max(0, n - 8)
        assert got.shape == (n - k0, 6)
        for i, k in enumerate(range(k0, n)):
            assert got[i, 2] == host[e].rewards[k] and got[i, 4] == all_len[e][k]
            assert math.isclose(got[i, 0], host[e].efficiency_scores[k], rel_tol=1e-12, abs_tol=0.0)
            assert math.isclose(got[i, 1], host[e].effort_scores[k], rel_tol=1e-12, abs_tol=0.0)
        assert math.isclose(sums[e, 2], sum(host[e].rewards), rel_tol=1e-12, abs_tol=1e-9)
        assert math.isclose(sums[e, 0], sum(host[e].efficiency_scores), rel_tol=1e-12) and math.isclose(sums[e, 1], sum(host[e].effort_scores), rel_tol=1e-12)
        assert sums[e, 4] == sum(all_len[e])
    env.check_errors()
    env.close()


def test_random_start_matches_oracle_and_redraws_every_episode():
    """cfg.random_start (ship_ice_env.py:201-203): start = (1 + u * (start_x_range - 1), 1, pi/2) drawn per (env, episode) from the counter
    RNG, then the 1000 settle sub-steps with the ship there; bit-identical to the oracle reset with that start."""
    from benchpush_amd.envs.ship_ice import default_trials
    from oracle.oracle import OracleShipIce
    E, T = 3, 2
    trials = default_trials(0.3, T, base_seed=5, goal_y=2.4)
    env = _mk(E, 0.3, trials, goal_y=2.4, random_start=True, start_x_range=7, start_seed=99)
    c = env.cfg
    orcs = [OracleShipIce(env.params, c.ship.vertices, c.ship.head, c.ship.tail) for _ in range(E)]
    eps = [0] * E

    def start_of(e):
        u = env.start_uniform(e, eps[e])
        assert 0.0 <= u < 1.0
        return (1 + u * (7 - 1), 1.0, np.pi / 2)

    obs, info = env.reset()
    starts = set()
    for e in range(E):
        oo, _ = orcs[e].reset(trials[(e + eps[e]) % T], start=start_of(e))
        assert np.array_equal(obs[e].cpu().numpy(), oo)
        starts.add(round(start_of(e)[0], 9))
    rng = np.random.default_rng(1)
    nreset = 0
    for t in range(24):
        a = rng.uniform(-0.3, 0.3, E)
        obs, rew, term, _, info = env.step(torch.from_numpy(a))
        bs = env.body_state().cpu().numpy()
        tm = term.cpu().numpy().astype(bool)
        for e in range(E):
            oo, orr, ot, _ = orcs[e].step(float(a[e]))
            nb = len(orcs[e].bodies())
            assert np.array_equal(bs[e, :nb], orcs[e].bodies()) and float(rew[e]) == orr and bool(tm[e]) == ot
            assert np.array_equal(obs[e].cpu().numpy(), oo)
        if tm.any():
            obs, _ = env.reset(term)
            for e in range(E):
                if tm[e]:
                    eps[e] += 1
                    nreset += 1
                    oo, _ = orcs[e].reset(trials[(e + eps[e]) % T], start=start_of(e))
                    assert np.array_equal(obs[e].cpu().numpy(), oo)
                    starts.add(round(start_of(e)[0], 9))
    assert nreset >= 2 and len(starts) >= 4
    env.check_errors()
    env.close()


@pytest.mark.parametrize("E,conc,steps,ntrials", [(4096, 0.3, 300, 50), (4096, 0.5, 300, 100)])
def test_whole_episode_soak_no_capacity_flags(E, conc, steps, ntrials):
    """Episodes run to their 300-step limit (environments/__init__.py:6) with auto-reset; the in-kernel capacities (arbiter slots,
    velocity slots, neighbour lists, colours, query buffers) must never be hit deep in an episode -- check_errors() every 25 steps.  The second case is
    the per-GPU shard of BASELINE.json configs[4] (C5: 32 768 envs at 50 % over 8 GPUs = 4096 envs x 50 % per GPU, 100 trials), whole episodes."""
    from benchpush_amd.envs.ship_ice import default_trials
    trials = default_trials(conc, ntrials, base_seed=0)
    env = _mk(E, conc, trials)
    env.reset()
    g = torch.Generator(device=env.device)
    g.manual_seed(2024)
    age = torch.zeros(E, dtype=torch.int64, device=env.device)
    done = 0
    for t in range(steps):
        a = (torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1) * 0.5  # gentler turns: longer episodes
        _, rew, term, _, info = env.step(a)
        age += 1
        m = term.bool() | (age >= 300)
        done += int(m.sum().item())
        age[m] = 0
        env.reset(m)
        if t % 25 == 24:
            env.check_errors()
            assert bool(torch.isfinite(rew).all()) and bool(torch.isfinite(info).all())
    env.check_errors()
    rows, cnt = env.episode_metrics()
    assert int(cnt.sum().item()) == done and done > 0
    fin = rows[cnt > 0]
    assert bool(((fin[:, 0] >= 0) & (fin[:, 0] <= 1.0 + 1e-9)).all()) and bool(((fin[:, 1] > 0) & (fin[:, 1] <= 1.0)).all())
    env.close()


def test_last_rank_shard_of_c5_matches_oracle_on_sampled_envs():
    """BASELINE.json configs[4] (C5) as its LAST rank sees it: 4096 envs at 50 %, `env_id_offset = 7 * 4096` (the other soaks run rank 0's shard).
    Env e of the shard plays trial (7 * 4096 + e + episode) % T; 8 envs spread over the shard are compared bit for bit with the oracle for 20 steps
    (body state, reward, termination, an observation every 5 steps, resets included), the whole shard is soaked for 60 steps with check_errors()."""
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
    from oracle.oracle import OracleShipIce
    E, T, OFF = 4096, 100, 7 * 4096
    trials = default_trials(0.5, T, base_seed=0)
    env = BatchedShipIceEnv(E, cfg={"concentration": 0.5}, trials=trials, device="cuda:0", env_id_offset=OFF)
    c = env.cfg
    sample = [0, 1, 511, 1024, 2049, 3000, 4094, 4095]
    orcs = {e: OracleShipIce(env.params, c.ship.vertices, c.ship.head, c.ship.tail) for e in sample}
    eps = {e: 0 for e in sample}
    obs, _ = env.reset()
    for e in sample:
        oo, _ = orcs[e].reset(trials[(OFF + e) % T])
        assert np.array_equal(obs[e].cpu().numpy(), oo), e
    g = torch.Generator(device=env.device)
    g.manual_seed(77)
    for t in range(60):
        a = (torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1) * 0.6
        obs, rew, term, _, info = env.step(a)
        if t < 20:
            bs = env.body_state()
            ah = a.cpu().numpy()
            for e in sample:
                oo, orr, ot, _ = orcs[e].step(float(ah[e]), observe=(t % 5 == 0))
                nb = len(orcs[e].bodies())
                assert np.array_equal(bs[e, :nb].cpu().numpy(), orcs[e].bodies()), (t, e)
                assert float(rew[e]) == orr and bool(term[e]) == ot, (t, e)
                if t % 5 == 0:
                    assert np.array_equal(obs[e].cpu().numpy(), oo), (t, e)
        obs2, _ = env.reset(term)
        if t < 20:
            tm = term.cpu().numpy().astype(bool)
            for e in sample:
                if tm[e]:
                    eps[e] += 1
                    oo, _ = orcs[e].reset(trials[(OFF + e + eps[e]) % T])
                    assert np.array_equal(obs2[e].cpu().numpy(), oo), (t, e)
        if t % 20 == 19:
            env.check_errors()
            assert bool(torch.isfinite(rew).all()) and bool(torch.isfinite(info).all())
    env.check_errors()
    env.close()


@pytest.mark.parametrize("E,conc,steps", [(8, 0.3, 300), (4, 0.5, 120)])
def test_sampled_envs_bit_exact_over_300_steps(E, conc, steps):
    """8 envs x 300 steps at 30 % (and 4 x 120 at 50 %, the concentration of config C5) against the oracle -- the parity suite's other
    cases stop at 40 steps: body state, rewards, termination and observations stay identical through whole episodes, resets included."""
    from benchpush_amd.envs.ship_ice import default_trials
    from oracle.oracle import OracleShipIce
    T = 4
    trials = default_trials(conc, T, base_seed=40)
    env = _mk(E, conc, trials)
    c = env.cfg
    orcs = [OracleShipIce(env.params, c.ship.vertices, c.ship.head, c.ship.tail) for _ in range(E)]
    obs, _ = env.reset()
    eps = [0] * E
    for e in range(E):
        oo, _ = orcs[e].reset(trials[e % T])
        assert np.array_equal(obs[e].cpu().numpy(), oo)
    rng = np.random.default_rng(8)
    age = np.zeros(E, int)
    for t in range(steps):
        a = rng.uniform(-1, 1, E) * 0.35
        obs, rew, term, _, info = env.step(torch.from_numpy(a))
        bs = env.body_state().cpu().numpy()
        tm = term.cpu().numpy().astype(bool)
        ob = obs.cpu().numpy() if t % 10 == 0 else None
        for e in range(E):
            oo, orr, ot, _ = orcs[e].step(float(a[e]), observe=ob is not None)
            nb = len(orcs[e].bodies())
            assert np.array_equal(bs[e, :nb], orcs[e].bodies()), (t, e)
            assert float(rew[e]) == orr and bool(tm[e]) == ot, (t, e)
            if ob is not None:
                assert np.array_equal(ob[e], oo), (t, e)
        age += 1
        m = tm | (age >= 300)
        if m.any():
            obs, _ = env.reset(torch.from_numpy(m.astype(np.uint8)))
            for e in range(E):
                if m[e]:
                    eps[e] += 1
                    age[e] = 0
                    oo, _ = orcs[e].reset(trials[(e + eps[e]) % T])
                    assert np.array_equal(obs[e].cpu().numpy(), oo)
    env.check_errors()
    env.close()


def test_reference_schema_pickle_through_the_hip_env(tmp_path, monkeypatch):
    """an experiment file with the reference's name and pickle schema ({'meta_data', 'exp': {conc: {trial: {'goal',
    'ship_state', 'obstacles'}}}}, ship_ice_env.py:76-80,188-198), written by the restated generate_rand_exp pipeline, is picked up through
    $BENCHPUSH_ICE_DIR by the gym-shaped ShipIceEnv and by the batched env, and steps bit-identically to the oracle on its trials."""
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, ShipIceEnv, experiment_file, resolve_trials
    from benchpush_amd.config import default_cfg, merge_user_cfg
    from benchpush_amd.ice_field_generator import generate_rand_exp
    from benchpush_amd.scenario import load_experiment
    from oracle.oracle import OracleShipIce
    path = experiment_file(0.2, str(tmp_path))
    generate_rand_exp(0.2, max_trials=3, filename=path, seed=5, goal=(0, 2.5))
    monkeypatch.setenv("BENCHPUSH_ICE_DIR", str(tmp_path))
    cfg = merge_user_cfg(default_cfg("ship_ice"), {"concentration": 0.2, "goal_y": 2.5})
    trials = resolve_trials(cfg)
    exp = load_experiment(path, 0.2)
    assert len(trials) == 3 and all(np.array_equal(trials[k]["obstacles"][0]["vertices"], exp[k]["obstacles"][0]["vertices"]) for k in range(3))
    env = BatchedShipIceEnv(3, cfg={"concentration": 0.2, "goal_y": 2.5}, device="cuda:0")        # trials come from the file
    assert len(env.trials) == 3
    c = env.cfg
    orcs = [OracleShipIce(env.params, c.ship.vertices, c.ship.head, c.ship.tail) for _ in range(3)]
    obs, _ = env.reset()
    for e in range(3):
        oo, _ = orcs[e].reset(trials[e])
        assert np.array_equal(obs[e].cpu().numpy(), oo)
    rng = np.random.default_rng(2)
    for t in range(10):
        a = rng.uniform(-0.5, 0.5, 3)
        obs, rew, term, _, info = env.step(torch.from_numpy(a))
        bs = env.body_state().cpu().numpy()
        for e in range(3):
            oo, orr, ot, _ = orcs[e].step(float(a[e]))
            nb = len(orcs[e].bodies())
            assert np.array_equal(bs[e, :nb], orcs[e].bodies()) and float(rew[e]) == orr and bool(term[e]) == ot
            assert np.array_equal(obs[e].cpu().numpy(), oo)
    env.check_errors()
    env.close()
    g = ShipIceEnv(cfg={"concentration": 0.2, "goal_y": 2.5}, device="cuda:0")
    o0, info0 = g.reset()
    oo, _ = OracleShipIce(env.params, c.ship.vertices, c.ship.head, c.ship.tail).reset(trials[0])
    assert np.array_equal(o0, oo) and len(info0["obs"]) == len(trials[0]["obstacles"])
    g.close()
