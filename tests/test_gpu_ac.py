"""area-clearing-v0: GPU (C ABI, bp_bd_config.task = 1) vs oracle, bit-exact on bodies, info, rewards, flags and observations."""
import numpy as np
import pytest
import torch

from benchpush_amd import area_clearing_scenario as A
from benchpush_amd.config import default_cfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("layout,atype", [("clear_env", "heading"), ("walled_env_with_columns", "position"), ("clear_env_small", "velocity")])
def test_area_clearing_matches_oracle(layout, atype):
    from benchpush_amd.envs.area_clearing import BatchedAreaClearingEnv
    from oracle.oracle_bd import AC_INFO_KEYS, OracleAreaClearing
    cfg = default_cfg("area_clearing")
    cfg.env = layout
    cfg.agent.action_type = atype
    trials = A.generate_trials(cfg, 4)
    E = 4
    env = BatchedAreaClearingEnv(E, cfg={"env": layout, "agent": {"action_type": atype}}, trials=trials)
    oracles = []
    for e in range(E):
        o = OracleAreaClearing(A.area_clearing_physics_params(cfg), A.area_clearing_params(cfg), cfg)
        o.reset(trials[e], observe=False)
        oracles.append(o)
    m, om = env.maps(0), oracles[0].maps()
    d = m["dims"]
    win = (slice(int(d[4]), int(d[4]) + int(d[2])), slice(int(d[5]), int(d[5]) + int(d[3])))
    assert np.array_equal(m["cspace"], om["cspace"][win]) and np.array_equal(m["recept"], om["recept"][win])
    obs, _ = env.reset()
    torch.cuda.synchronize()
    assert np.array_equal(obs.cpu().numpy(), np.stack([o.observe() for o in oracles]))
    rng = np.random.RandomState(3)
    for t in range(5):
        if atype == "velocity":
            a = rng.uniform(-1, 1, (E, 2))
        elif atype == "position":
            a = rng.randint(0, 224 * 224, E).astype(np.float64)
        else:
            a = rng.uniform(-1, 1, E)
        obs, rew, term, trunc, info = env.step(torch.tensor(a))
        torch.cuda.synchronize()
        res = [o.step(a[e]) for e, o in enumerate(oracles)]
        assert np.array_equal(info.cpu().numpy(), np.array([[r[4][k] for k in AC_INFO_KEYS] for r in res])), t
        assert np.array_equal(rew.cpu().numpy(), np.array([r[1] for r in res])), t
        assert np.array_equal(term.cpu().numpy().astype(bool), np.array([r[2] for r in res]))
        assert np.array_equal(trunc.cpu().numpy().astype(bool), np.array([r[3] for r in res]))
        assert np.array_equal(obs.cpu().numpy(), np.stack([r[0] for r in res])), t
        st = env.body_state().cpu().numpy()
        for e, o in enumerate(oracles):
            n = 6 + env.nbox
            assert np.array_equal(st[e, :n], o.shape_states()[:n]), (t, e)
    env.check_errors()
    env.close()


def test_area_clearing_nonzero_damping_matches_oracle():
    """cfg.sim.damping = 0.8 on area-clearing-v0 (the same DAMP instantiation as box-delivery): 3 envs x 4 heading steps against the oracle."""
    from benchpush_amd.envs.area_clearing import BatchedAreaClearingEnv
    from oracle.oracle_bd import AC_INFO_KEYS, OracleAreaClearing
    cfg = default_cfg("area_clearing")
    cfg.sim.damping = 0.8
    E = 3
    trials = A.generate_trials(cfg, E)
    env = BatchedAreaClearingEnv(E, cfg={"sim": {"damping": 0.8}}, trials=trials)
    oracles = []
    for e in range(E):
        o = OracleAreaClearing(A.area_clearing_physics_params(cfg), A.area_clearing_params(cfg), cfg)
        o.reset(trials[e], observe=False)
        oracles.append(o)
    obs, _ = env.reset()
    torch.cuda.synchronize()
    assert np.array_equal(obs.cpu().numpy(), np.stack([o.observe() for o in oracles]))
    rng = np.random.RandomState(5)
    for t in range(4):
        a = rng.uniform(-1, 1, E)
        obs, rew, term, trunc, info = env.step(torch.tensor(a))
        torch.cuda.synchronize()
        res = [o.step(a[e]) for e, o in enumerate(oracles)]
        assert np.array_equal(info.cpu().numpy(), np.array([[r[4][k] for k in AC_INFO_KEYS] for r in res])), t
        assert np.array_equal(rew.cpu().numpy(), np.array([r[1] for r in res])), t
        assert np.array_equal(obs.cpu().numpy(), np.stack([r[0] for r in res])), t
        st = env.body_state().cpu().numpy()
        for e, o in enumerate(oracles):
            n = 6 + env.nbox
            assert np.array_equal(st[e, :n], o.shape_states()[:n]), (t, e)
    env.check_errors()
    env.close()


@pytest.mark.parametrize("extra_wall", [[[-7.0, 0.0], [-3.0, 0.0]], [[-4.95, 0.0], [-3.0, 0.0]]], ids=["wall_across_the_edge", "wall_end_cap_reaches_the_edge"])
def test_layout_whose_wall_cuts_the_clearance_boundary_matches_oracle(extra_wall):
    """VERDICT r3 item 9 / r4 item 9 (area_clearing.py:225-262,1122-1140): walled_env plus a wall across the left boundary edge, or one whose rounded end (the
    cap of its 0.1 m buffer) reaches it -- five boundary goal lines, 50 goal points; maps, first observation and five steps (goal-distance rewards, channel 3)
    against the oracle."""
    from benchpush_amd.envs.area_clearing import BatchedAreaClearingEnv
    from oracle.oracle_bd import AC_INFO_KEYS, OracleAreaClearing
    walls = [[[-6.0, -6.0], [-6.0, 6.0]], [[6.0, 6.0], [6.0, -6.0]], extra_wall]
    cfg = default_cfg("area_clearing")
    cfg.env = "walled_env"
    cfg.envs.walled_env.walls = walls
    assert A.goal_points(cfg).shape == (50, 2)
    E = 3
    trials = A.generate_trials(cfg, E)
    env = BatchedAreaClearingEnv(E, cfg={"env": "walled_env", "envs": {"walled_env": cfg.envs.walled_env}}, trials=trials)   # one-level-deep merge: the layout entry is replaced
    assert len(env.goal_points) == 50
    oracles = []
    for e in range(E):
        o = OracleAreaClearing(A.area_clearing_physics_params(cfg), A.area_clearing_params(cfg), cfg)
        o.reset(trials[e], observe=False)
        oracles.append(o)
    m, om = env.maps(0), oracles[0].maps()
    d = m["dims"]
    win = (slice(int(d[4]), int(d[4]) + int(d[2])), slice(int(d[5]), int(d[5]) + int(d[3])))
    assert np.array_equal(m["cspace"], om["cspace"][win]) and np.array_equal(m["recept"], om["recept"][win])
    obs, _ = env.reset()
    torch.cuda.synchronize()
    assert np.array_equal(obs.cpu().numpy(), np.stack([o.observe() for o in oracles]))
    rng = np.random.RandomState(9)
    for t in range(5):
        a = rng.uniform(-1, 1, E)
        obs, rew, term, trunc, info = env.step(torch.tensor(a))
        torch.cuda.synchronize()
        res = [o.step(a[e]) for e, o in enumerate(oracles)]
        assert np.array_equal(info.cpu().numpy(), np.array([[r[4][k] for k in AC_INFO_KEYS] for r in res])), t
        assert np.array_equal(rew.cpu().numpy(), np.array([r[1] for r in res])), t
        assert np.array_equal(obs.cpu().numpy(), np.stack([r[0] for r in res])), t
    env.check_errors()
    env.close()


def test_area_clearing_two_pass_step_matches_oracle(monkeypatch):
    """area-clearing through the two-pass step with a budget that stops most env steps mid-path (BP_BD_BUDGET=200; an env step is ~950 sim steps): the
    resumed group and the group that finished in pass 0 against the oracle, bit for bit."""
    monkeypatch.setenv("BP_BD_BUDGET", "200")
    test_area_clearing_matches_oracle("clear_env", "heading")
    monkeypatch.setenv("BP_BD_BUDGET", "1000")
    test_area_clearing_matches_oracle("walled_env_with_columns", "position")


def test_deep_episodes_through_clearing_and_time_truncation():
    """30 env steps of 4 envs against the oracle with auto-reset (every step: bodies, info, reward, flags, observation; every reset: first
    observation): a hand-placed box next to the clearance boundary is pushed out (cleared reward, box_count; area_clearing.py:611-780) and the
    episodes are cut by t_max (lowered to 9 steps through the config, config.yaml:105) and restart on the next trials."""
    from benchpush_amd.envs.area_clearing import BatchedAreaClearingEnv
    from oracle.oracle_bd import AC_INFO_KEYS, OracleAreaClearing
    cfg = default_cfg("area_clearing")
    cfg.env = "clear_env_small"
    cfg.sim.t_max = 9
    gen = A.generate_trials(cfg, 3)
    tr = dict(gen[0])
    tr["start"] = np.array([2.2, 0.0, 0.0])                 # facing +x, 1.8 m from the boundary at x = 4
    tr["boxes"] = np.array(gen[0]["boxes"])
    tr["boxes"][0] = [3.2, 0.0, 0.0]                        # the box between robot and boundary
    trials = [tr, gen[1], gen[2]]
    E, T = 4, 3
    env = BatchedAreaClearingEnv(E, cfg={"env": "clear_env_small", "sim": {"t_max": 9}}, trials=trials)

    def mk(trial):
        o = OracleAreaClearing(A.area_clearing_physics_params(cfg), A.area_clearing_params(cfg), cfg)
        o.reset(trial, observe=False)
        return o

    oracles = [mk(trials[e % T]) for e in range(E)]
    episode = np.zeros(E, int)
    obs, _ = env.reset()
    rng = np.random.RandomState(17)
    cleared, resets, truncs = 0.0, 0, 0
    for t in range(30):
        a = rng.uniform(-1, 1, E)
        if t < 3:
            a[0] = a[3] = 0.0                               # straight ahead: envs 0 and 3 play the hand-placed trial
        obs, rew, term, trunc, info = env.step(torch.tensor(a))
        torch.cuda.synchronize()
        res = [o.step(a[e]) for e, o in enumerate(oracles)]
        assert np.array_equal(info.cpu().numpy(), np.array([[r[4][k] for k in AC_INFO_KEYS] for r in res])), t
        assert np.array_equal(rew.cpu().numpy(), np.array([r[1] for r in res])), t
        assert np.array_equal(term.cpu().numpy().astype(bool), np.array([r[2] for r in res])), t
        assert np.array_equal(trunc.cpu().numpy().astype(bool), np.array([r[3] for r in res])), t
        assert np.array_equal(obs.cpu().numpy(), np.stack([r[0] for r in res])), t
        st = env.body_state().cpu().numpy()
        for e, o in enumerate(oracles):
            n = 6 + env.nbox
            assert np.array_equal(st[e, :n], o.shape_states()[:n]), (t, e)
        cleared = max(cleared, max(r[4]["box_count"] for r in res))
        done = np.array([r[2] or r[3] for r in res])
        truncs += int(sum(r[3] for r in res))
        if done.any():
            obs, _ = env.reset(torch.from_numpy(done.astype(np.uint8)))
            torch.cuda.synchronize()
            for e in np.nonzero(done)[0]:
                episode[e] += 1
                oo = oracles[e].reset(trials[(e + episode[e]) % T])
                assert np.array_equal(obs[e].cpu().numpy(), oo), ("reset obs", t, e)
                resets += 1
    env.check_errors()
    env.close()
    assert cleared >= 1 and truncs >= 4 and resets >= 4, (cleared, truncs, resets)


def test_area_clearing_gym_adapter_and_metric():
    import benchpush_amd
    from benchpush_amd.metrics.task_driven_metric import TaskDrivenMetric
    g = benchpush_amd.make("area-clearing-v0", cfg={"env": "clear_env_small"}).unwrapped
    metric = TaskDrivenMetric(alg_name="random", robot_mass=g.cfg.agent.mass)
    obs, info = g.reset()
    metric.reset(info)
    assert obs.shape == (224, 224, 4) and obs.dtype == np.uint8
    assert {"state", "total_work", "obs", "box_count", "boundary", "walls", "static_obstacles", "goal_positions"} <= set(info)
    done = False
    for t in range(3):
        obs, r, term, trunc, info = g.step(np.array([0.2 * t - 0.3], np.float32))
        metric.update(info, r, eps_complete=(t == 2))
    assert len(info["obs"]) == 10 and len(info["box_completed_statuses"]) == 10 and isinstance(r, float)
    assert len(metric.effort_scores) == 1 and 0 < metric.effort_scores[0] <= 1.0 + 1e-9
    g.close()


def test_vec_env_and_on_device_policy_rollout():
    """SB3-shaped VecEnv over box-delivery (auto-reset, terminal_observation) and a torch policy consuming the device observations."""
    import torch.nn as nn
    from benchpush_amd.envs.vec_env import make_area_clearing_vec_env, make_box_delivery_vec_env
    venv = make_box_delivery_vec_env(3, num_trials=3)
    obs = venv.reset()
    assert obs.shape == (3, 224, 224, 4) and obs.dtype == np.uint8
    obs, rew, dones, infos = venv.step(np.array([[0.1], [-0.4], [0.8]], np.float32))
    assert obs.shape == (3, 224, 224, 4) and rew.shape == (3,) and len(infos) == 3 and "cumulative_distance" in infos[0]
    venv.close()
    venv = make_area_clearing_vec_env(4, num_trials=2, to_numpy=False) if False else make_area_clearing_vec_env(4, num_trials=2)
    venv.to_numpy = False
    obs = venv.reset()
    policy = nn.Sequential(nn.Flatten(), nn.Linear(224 * 224 * 4, 1), nn.Tanh()).to(obs.device)
    with torch.no_grad():
        for _ in range(2):
            act = policy(obs.float() / 255.0)
            obs, rew, dones, infos = venv.step(act)
    assert obs.is_cuda and obs.dtype == torch.uint8 and rew.is_cuda
    venv.close()
