"""Parity tests proper (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs.

Bar: bit-exact.  Physics state is binary64 on both sides with the same operation order, so body state, rewards and
info scalars are compared with ==, contact counters and termination flags exactly, observations byte for byte.
"""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(E, conc, trials, **kw):
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv
    return BatchedShipIceEnv(E, cfg=dict({"concentration": conc}, **kw), trials=trials, device="cuda:0")


def _oracles(env, n):
    from oracle.oracle import OracleShipIce
    c = env.cfg
    return [OracleShipIce(env.params, c.ship.vertices, c.ship.head, c.ship.tail) for _ in range(n)]


def _run_parity(E, conc, T, steps, seed, action_fn=None, **cfgkw):
    from benchpush_amd.envs.ship_ice import default_trials
    trials = default_trials(conc, T, base_seed=seed)
    env = _mk(E, conc, trials, **cfgkw)
    import os
    assert int(env.L.bp_pair_mode(env.h)) == int(os.environ.get("BP_PAIR", "0") or 0)   # a test that asks for the paired kernels must get them
    obs, info = env.reset()
    orcs = _oracles(env, E)
    eps = [0] * E
    for e, o in enumerate(orcs):
        oo, _ = o.reset(trials[e % T])
        assert np.array_equal(obs[e].cpu().numpy(), oo), ("reset obs", e)
    rng = np.random.default_rng(seed)
    ncontact = 0
    for t in range(steps):
        a = rng.uniform(-1, 1, E) if action_fn is None else np.array([action_fn(e, t) for e in range(E)], np.float64)
        a = a.astype(np.float32).astype(np.float64)
        obs, rew, term, trunc, info = env.step(torch.from_numpy(a))
        bs = env.body_state().cpu().numpy()
        nb = env.num_bodies()
        go, gi, gr, gt = obs.cpu().numpy(), info.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
        assert not trunc.any()
        for e, o in enumerate(orcs):
            oo, orr, ot, oi = o.step(float(a[e]))
            ob = o.bodies()
            assert nb[e] == len(ob)
            assert np.array_equal(bs[e, : nb[e]], ob), ("bodies", t, e)
            assert np.array_equal(go[e], oo), ("obs", t, e)
            assert np.array_equal(gi[e], np.array(list(oi.values()))), ("info", t, e)
            assert gr[e] == orr and bool(gt[e]) == ot, ("reward/term", t, e)
            ncontact = max(ncontact, int(oi["n_contact_pts"]))
        done = gt.astype(bool)
        if done.any():
            obs, info = env.reset(term)
            for e in np.nonzero(done)[0]:
                eps[e] += 1
                oo, _ = orcs[e].reset(trials[(e + eps[e]) % T])
                assert np.array_equal(obs[e].cpu().numpy(), oo), ("auto-reset obs", t, e)
    env.check_errors()
    return ncontact


def test_parity_30pct_with_contacts_and_autoreset():
    assert _run_parity(E=8, conc=0.3, T=3, steps=40, seed=0) > 1000


def test_parity_preemptive_scheduler(monkeypatch):
    """k_physics_step_sched (the default step kernel: chunks of 40 sub-steps, envs parked at a chunk boundary when another one is further behind and
    resumed by another workgroup) with other chunk sizes -- one that divides the 400 sub-steps, one that leaves a ragged last chunk -- and the
    one-wave-per-env kernel it replaces (BP_SCHED=0): all bit-identical to the oracle."""
    monkeypatch.setenv("BP_SCHED", "50")
    assert _run_parity(E=9, conc=0.3, T=3, steps=30, seed=5) > 500
    monkeypatch.setenv("BP_SCHED", "37")
    assert _run_parity(E=7, conc=0.5, T=2, steps=12, seed=21) > 100
    monkeypatch.setenv("BP_SCHED", "0")
    assert _run_parity(E=9, conc=0.3, T=3, steps=30, seed=5) > 500


@pytest.mark.parametrize("kind,conc", [("ship", 0.3), ("ship", 0.5), ("maze", None), ("box", None)])
def test_cached_plane_hints_never_change_results(kind, conc):
    """The hint word of a neighbour slot (winning plane and support vertex of both shapes, flags) only decides how much of the plane search is
    skipped: cached planes are evaluated exactly, every other plane is pruned by an upper bound or searched.  bp_debug_scramble_hints overwrites
    every written hint with random valid indices and random flags before every step; states, rewards, observations and info must stay bit-identical."""
    import ctypes as C
    if kind == "ship":
        from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
        trials = default_trials(conc, 5, base_seed=23)
        mk = lambda: BatchedShipIceEnv(E, cfg={"concentration": conc}, trials=trials, device="cuda:0")
    elif kind == "maze":
        from benchpush_amd.envs.maze_namo import BatchedMazeEnv
        mk = lambda: BatchedMazeEnv(E, cfg={"num_obstacles": 20}, num_layouts=5, base_seed=3, device="cuda:0")
    else:   # box-delivery: ~1000 sim steps of the same sub-step per env step
        from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
        mk = lambda: BatchedBoxDeliveryEnv(E, cfg={"env": {"obstacle_config": "small_columns"}}, num_trials=4, seed=5, device="cuda:0")
    E, steps = (64, 20) if kind != "box" else (16, 6)
    g = torch.Generator(device="cuda:0")
    g.manual_seed(17)
    acts = (torch.rand((steps, E), generator=g, device="cuda:0", dtype=torch.float64) * 2 - 1).float().double()

    def run(scramble):
        env = mk()
        env.reset()
        rsum = torch.zeros(E, dtype=torch.float64, device="cuda:0")
        for t in range(steps):
            if scramble:
                n = env.L.bp_debug_scramble_hints(env.h, C.c_uint64(1000 * scramble + t), None)
                assert n >= 0 and (t < 3 or n > E), n      # after a few steps every env has pairs with cached planes to overwrite
            _, rew, term, _, _ = env.step(acts[t])
            rsum += rew
            env.reset(term)
        env.check_errors()
        out = (env.body_state().clone(), rsum, env.obs.clone(), env.info.clone())
        env.close()
        return out

    ref = run(0)
    for sc in (1, 2):
        got = run(sc)
        for a, b in zip(ref, got):
            assert torch.equal(a, b), (kind, conc, sc)


@pytest.mark.parametrize("mask", [1, 2, 4, 8, 15])
def test_rarely_taken_narrow_phase_paths_match_oracle(monkeypatch, mask):
    """BP_DEBUG_PATHS forces the fallbacks of the narrow phase that the fast paths normally shadow: 1 = no candidate cache, 2 = bound rounds through the
    sequential loop that flushes the query buffer, 4 = cached planes through the eight-lane support query instead of the three-dot certificate,
    8 = manifold support vertices through the support query instead of the certificate of the winning edge.  Each is bit-identical to the oracle."""
    from benchpush_amd import _lib
    from benchpush_amd.build import DBG_LIB_PATH, build_debug_paths
    build_debug_paths()                                   # -DBP_DEBUG_PATHS twin of the library (built by __graft_entry__.build(); up to date -> no-op)
    monkeypatch.setattr(_lib, "_lib", None)               # load the twin for this test only; monkeypatch restores the product library afterwards
    monkeypatch.setenv("BP_PROF", "1")
    monkeypatch.setenv("BP_PROF_LIB", DBG_LIB_PATH)
    monkeypatch.setenv("BP_DEBUG_PATHS", str(mask))
    assert _run_parity(E=6, conc=0.3, T=3, steps=24, seed=9) > 300
    if mask in (2, 15):
        assert _run_parity(E=4, conc=0.5, T=2, steps=10, seed=4) > 100


def test_dispatch_order_hint_never_changes_results():
    """bp_set_step_cost_hint (the dispatch order of the step kernel) with random and with reversed hints: same states, rewards and observations."""
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
    trials = default_trials(0.3, 6, base_seed=11)
    E, steps = 96, 25
    g = torch.Generator(device="cuda:0")
    g.manual_seed(3)
    acts = (torch.rand((steps, E), generator=g, device="cuda:0", dtype=torch.float64) * 2 - 1).float().double()

    def run(mode):
        env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials, device="cuda:0")
        assert env.sched_chunk() == 40                      # the preemptive scheduler is the default step kernel
        env.reset()
        rng = np.random.default_rng(5)
        rsum = torch.zeros(E, dtype=torch.float64, device="cuda:0")
        for t in range(steps):
            if mode == "random":
                env.set_cost_hint(rng.integers(0, 1 << 20, E))
            elif mode == "reversed":
                env.set_cost_hint((1 << 30) - env.step_cycles().astype(np.int64) // 256)
            _, rew, term, _, _ = env.step(acts[t])
            rsum += rew
            env.reset(term)
        env.check_errors()
        out = (env.body_state().clone(), rsum, env.obs.clone(), env.info.clone())
        env.close()
        return out

    ref = run(None)
    for mode in ("random", "reversed"):
        got = run(mode)
        for a, b in zip(ref, got):
            assert torch.equal(a, b), mode


@pytest.mark.parametrize("damping", [0.9, 0.5])
def test_parity_nonzero_damping(damping):
    """`space.damping = cfg.sim.damping` (ship_ice_env.py:120) with a value other than the shipped 0: the handle runs the generic instantiation
    (k_physics_step_damp / k_physics_reset_damp: velocities scaled by damping^dt, floes coast after the push, moving list from the velocity slots);
    4 envs x 25 steps against the oracle, bit for bit -- reset (1000 settle sub-steps with damping), contacts, auto-reset."""
    n = _run_parity(E=4, conc=0.3, T=2, steps=25, seed=21, sim={"damping": damping})
    assert n > 100


def test_damping_is_a_real_knob_and_zero_matches_the_default():
    """damping 0.9 changes the trajectory of pushed floes (they keep moving), damping 0 given explicitly is the default path bit for bit."""
    from benchpush_amd.envs.ship_ice import default_trials
    trials = default_trials(0.3, 2, base_seed=21)
    outs = []
    for kw in ({}, {"sim": {"damping": 0.0}}, {"sim": {"damping": 0.9}}):
        env = _mk(4, 0.3, trials, **kw)
        env.reset()
        for t in range(25):   # the ship reaches the first floes after ~10 steps
            env.step(torch.full((4,), 0.25 * ((t % 3) - 1), dtype=torch.float64))
        outs.append(env.body_state().cpu().numpy())
        env.check_errors()
        env.close()
    assert np.array_equal(outs[0], outs[1])
    assert not np.array_equal(outs[0], outs[2])


def test_parity_negative_zero_yaw_command():
    """A ship commanded with -0.0 carries w = -0 (an infinite-mass body with a negative-zero velocity component): the register-resident solve of single-colour
    sub-steps must not be taken then (x + (+0) != x for x = -0; substep(), step 6d) -- the slot loop runs instead and the results equal the oracle's, as they do
    for +0.0 where the register path is taken."""
    n = _run_parity(E=4, conc=0.3, T=2, steps=30, seed=3, action_fn=lambda e, t: -0.0 if (e % 2) else 0.0)
    assert n > 100


def test_parity_50pct_dense_field():
    assert _run_parity(E=4, conc=0.5, T=2, steps=12, seed=21) > 100


def test_parity_10pct_plumbing_config():
    # BASELINE.json configs[0]: 1 env, 10 % concentration, action 0
    _run_parity(E=1, conc=0.1, T=1, steps=12, seed=3, action_fn=lambda e, t: 0.0)


def test_parity_boundary_and_yaw_edges():
    # hard-over rudder: reaches the yaw limit, then the channel boundary (-50, termination without success)
    _run_parity(E=2, conc=0.1, T=2, steps=30, seed=5, action_fn=lambda e, t: 1.0 if e == 0 else -1.0)


def test_parity_goal_variant():
    _run_parity(E=2, conc=0.2, T=2, steps=8, seed=8, goal_y=19)


def _edge_trials():
    """Ragged / extreme scenario tables: an empty channel, one 20-vertex floe dead ahead, a triangle, a zero-area floe that the reference's
    loader drops (ship_ice_env.py:205-209 via sim_utils), and a dense field -- all in one handle, so floe counts differ per trial."""
    from benchpush_amd.scenario import generate_ice_field
    ang = np.linspace(0, 2 * np.pi, 20, endpoint=False)
    big = np.stack([6.0 + 0.7 * np.cos(ang), 2.6 + 0.7 * np.sin(ang)], 1)
    tri = np.array([[5.2, 2.2], [6.8, 2.3], [6.1, 3.4]])
    flat = np.array([[3.0, 5.0], [4.0, 5.0], [5.0, 5.0]])          # collinear: zero area
    mk = lambda v: {"vertices": v, "centre": tuple(v.mean(0)), "radius": float(np.abs(v - v.mean(0)).max())}
    start = (6.0, 1.0, np.pi / 2)
    return [{"goal": (0, 9.0), "ship_state": start, "obstacles": []},
            {"goal": (0, 9.0), "ship_state": start, "obstacles": [mk(big)]},
            {"goal": (0, 9.0), "ship_state": start, "obstacles": [mk(tri), mk(flat)]},
            generate_ice_field(0.5, 99, min_r=0.4, max_r=0.58)]


def test_parity_empty_ragged_and_extreme_scenarios():
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv
    trials = _edge_trials()
    E = 4
    env = BatchedShipIceEnv(E, cfg={"concentration": 0.5}, trials=trials)
    obs, info = env.reset()
    orcs = _oracles(env, E)
    for e, o in enumerate(orcs):
        oo, _ = o.reset(trials[e])
        assert np.array_equal(obs[e].cpu().numpy(), oo), ("reset obs", e)
    nb = env.num_bodies()
    assert nb[0] == 1 and nb[1] == 2 and nb[2] == 2 and nb[3] > 200      # the zero-area floe is dropped, like the reference does
    rng = np.random.default_rng(17)
    done_seen = np.zeros(E, bool)
    for t in range(36):
        a = np.where(np.arange(E) < 3, 0.0, rng.uniform(-1, 1, E)).astype(np.float32).astype(np.float64)
        obs, rew, term, trunc, info = env.step(torch.from_numpy(a))
        bs = env.body_state().cpu().numpy()
        for e, o in enumerate(orcs):
            if done_seen[e]:
                continue
            oo, orr, ot, oi = o.step(float(a[e]))
            ob = o.bodies()
            assert np.array_equal(bs[e, : len(ob)], ob), ("bodies", t, e)
            assert np.array_equal(obs[e].cpu().numpy(), oo), ("obs", t, e)
            assert float(rew[e]) == orr and bool(term[e]) == ot, ("reward/term", t, e)
            assert np.array_equal(info[e].cpu().numpy(), np.array(list(oi.values()))), ("info", t, e)
            done_seen[e] |= ot
    assert done_seen[0]                      # the empty channel is crossed: 8 m at 0.3 m/s * 0.8 s per step = 34 steps
    env.check_errors()
    env.close()


def test_world_polys_and_low_dim_match_oracle():
    from benchpush_amd.envs.ship_ice import default_trials
    trials = default_trials(0.2, 1, base_seed=4)
    env = _mk(1, 0.2, trials)
    env.reset()
    o = _oracles(env, 1)[0]
    o.reset(trials[0])
    for t in range(12):
        env.step(torch.tensor([0.2]))
        o.step(float(np.float32(0.2)))
    v, c = env.world_polys()
    ov, oc = o.world_polys()
    nb = len(oc)
    assert np.array_equal(c[0, :nb].cpu().numpy(), oc)
    assert np.array_equal(v[0, :nb].cpu().numpy(), ov[:, :20])
    from oracle.oracle import poly_centroid
    ld = env.low_dim_obs()[0, : nb - 1].cpu().numpy()
    for i in range(1, nb):
        assert np.array_equal(ld[i - 1], np.abs(poly_centroid(ov[i, : oc[i]])))


def test_sharding_invariance_and_determinism():
    """Results depend on the global env id only: one handle of 8 envs == two handles of 4 with env_id_offset 0 / 4."""
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
    trials = default_trials(0.3, 5, base_seed=2)
    acts = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (10, 8)).astype(np.float32).astype(np.float64))

    def run(E, off):
        env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials, device="cuda:0", env_id_offset=off)
        env.reset()
        out = []
        for t in range(10):
            obs, rew, term, _, info = env.step(acts[t, off:off + E])
            out.append((obs.cpu().numpy().copy(), rew.cpu().numpy().copy(), info.cpu().numpy().copy()))
            env.reset(term)
        return out

    full, lo, hi, again = run(8, 0), run(4, 0), run(4, 4), run(8, 0)
    for t in range(10):
        for k in range(3):
            assert np.array_equal(full[t][k], np.concatenate([lo[t][k], hi[t][k]]))
            assert np.array_equal(full[t][k], again[t][k])


def test_reset_from_settled_template_equals_in_place_settle():
    """bp_reset copies a per-trial template settled at load time; re-running the 1000 settle sub-steps in place gives
    the same bits (state, observation, info), also for later episodes and after stepping."""
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
    trials = default_trials(0.4, 3, base_seed=13)
    outs = []
    for resettle in (False, True):
        env = BatchedShipIceEnv(6, cfg={"concentration": 0.4}, trials=trials, device="cuda:0")
        env.set_resettle(resettle)
        rec = []
        obs, info = env.reset()
        rec.append((obs.cpu().numpy().copy(), info.cpu().numpy().copy(), env.body_state().cpu().numpy().copy()))
        for t in range(3):
            obs, rew, term, _, info = env.step(torch.full((6,), 0.3, dtype=torch.float64))
        m = torch.tensor([1, 0, 1, 0, 0, 1], dtype=torch.uint8)
        obs, info = env.reset(m)
        rec.append((obs.cpu().numpy().copy(), info.cpu().numpy().copy(), env.body_state().cpu().numpy().copy()))
        obs, rew, term, _, info = env.step(torch.full((6,), -0.2, dtype=torch.float64))
        rec.append((obs.cpu().numpy().copy(), info.cpu().numpy().copy(), env.body_state().cpu().numpy().copy()))
        env.check_errors()
        outs.append(rec)
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_full_size_properties_4096_envs():
    """BASELINE.json configs[1] size: properties that need no oracle run."""
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
    trials = default_trials(0.3, 16, base_seed=0)
    E = 4096
    env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials, device="cuda:0")
    obs, info = env.reset()
    x0 = info[:, 0].clone()
    # envs that share a trial and an action sequence must stay bit-identical (trial = global id % 16)
    g = torch.Generator(device="cuda:0")
    g.manual_seed(0)
    base = (torch.rand((6, 16), generator=g, device="cuda:0", dtype=torch.float64) * 2 - 1).float().double()
    prev_tw = torch.zeros(E, dtype=torch.float64, device="cuda:0")
    for t in range(6):
        a = base[t].repeat(E // 16)
        obs, rew, term, trunc, info = env.step(a)
        assert torch.equal(obs.view(E // 16, 16, -1), obs.view(E // 16, 16, -1)[0:1].expand(E // 16, -1, -1))
        assert torch.equal(rew.view(-1, 16), rew.view(-1, 16)[0:1].expand(E // 16, -1))
        assert (info[:, 4] >= 0).all() and (info[:, 3] >= prev_tw).all()           # work >= 0, total_work monotone
        prev_tw = info[:, 3].clone()
        assert torch.allclose(info[:, 1], torch.full_like(info[:, 1], 1.0) + 0.24 * (t + 1), atol=0.24 * (t + 1))
        assert not trunc.any() and not term.any()
        ch0 = obs[:, 0]
        assert ((ch0 == 0) | (ch0 == 127) | (ch0 == 255)).all()                      # footprint channel values
        assert ((obs[:, 3] == 0) | (obs[:, 3] == 255)).all()                         # occupancy is binary
        assert ((obs[:, 2] == 0) | (obs[:, 2] == 127) | (obs[:, 2] == 255)).all()
        assert (obs[:, 2] == 255).sum(dim=(1, 2)).eq(1).all()                        # exactly one head pixel
    env.check_errors()
    assert torch.isfinite(info).all()
    assert x0.min() >= 1.0 and x0.max() <= 11.0


def test_step_kernel_variants_agree_at_full_size(monkeypatch):
    """4096 envs x 40 steps with auto-reset (deep enough for the cost-sorted dispatch order and for thousands of parked envs): the preemptive
    scheduler (the default on resident wavefronts, a ragged chunk size, and the dispatcher-driven kernel) leaves every env in exactly the state of the one-env-per-wavefront kernel (BP_SCHED=0) --
    body state, rewards, termination, observations, episode metrics."""
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
    trials = default_trials(0.3, 24, base_seed=3)
    E, steps = 4096, 40
    g = torch.Generator(device="cuda:0")
    g.manual_seed(5)
    acts = (torch.rand((steps, E), generator=g, device="cuda:0", dtype=torch.float64) * 2 - 1).float().double()

    def run(env_vars):
        for k in ("BP_SCHED", "BP_SCHED_PERSIST", "BP_SCHED_DYNPRIO"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env_vars.items():
            monkeypatch.setenv(k, v)
        env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials, device="cuda:0")
        # the default runs on resident wavefronts (one workgroup per wave slot), BP_SCHED_PERSIST=0 on one workgroup per task from the hardware dispatcher
        if env_vars.get("BP_SCHED") != "0":
            assert (int(env.L.bp_sched_resident(env.h)) > 0) == (env_vars.get("BP_SCHED_PERSIST") != "0")
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

    ref = run({"BP_SCHED": "0"})                               # one env per wavefront for the whole step
    assert ref[5] > 1000                                       # episodes did end and restart inside the window
    # {} = the default: the preemptive scheduler (k_physics_step_sched), here with thousands of envs parked and resumed on other CUs / XCDs
    # ... on resident wavefronts with pace-based issue priorities; the same without either (the dispatcher-driven kernel, static priority classes)
    for variant in ({}, {"BP_SCHED": "37"}, {"BP_SCHED_PERSIST": "0", "BP_SCHED_DYNPRIO": "0"}):
        got = run(variant)
        for a, b in zip(ref[:5], got[:5]):
            assert torch.equal(a, b), variant
        assert got[5] == ref[5]


def test_scheduler_fault_is_finished_by_the_completion_launch(monkeypatch):
    """Scheduler fault path (bp_kernels.hpp: sched_body): with the test hook BP_SCHED_DEBUG_DROP=1 env 1 is parked after its first chunk and its queue
    item is dropped, so the scheduled launch cannot finish it; the pollers leave on the (short) watchdog and the completion launch that follows every
    scheduled launch resumes the env at its chunk boundary.  Results equal the unscheduled kernel's bit for bit, the fault is reported as a warning."""
    from benchpush_amd import _lib
    from benchpush_amd.build import DBG_LIB_PATH, build_debug_paths
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
    # the fault-injection hook is compiled into the diagnostic twin of the library only: the product library ignores the variable,
    # which the last run below checks
    build_debug_paths()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("BP_PROF", "1")
    monkeypatch.setenv("BP_PROF_LIB", DBG_LIB_PATH)
    trials = default_trials(0.3, 3, base_seed=5)
    E, steps = 9, 12
    g = torch.Generator(device="cuda:0")
    g.manual_seed(11)
    acts = (torch.rand((steps, E), generator=g, device="cuda:0", dtype=torch.float64) * 2 - 1).float().double()

    def run(env_vars):
        for k in ("BP_SCHED", "BP_SCHED_DEBUG_DROP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env_vars.items():
            monkeypatch.setenv(k, v)
        env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials, device="cuda:0")
        env.reset()
        rews = []
        for t in range(steps):
            obs, rew, term, _, info = env.step(acts[t])
            rews.append(rew.clone())
            env.reset(term)
        env.check_errors()
        out = (env.body_state().clone(), torch.stack(rews), env.obs.clone(), env.info.clone(), env.sched_warnings())
        env.close()
        return out

    ref = run({"BP_SCHED": "0"})
    got = run({"BP_SCHED_DEBUG_DROP": "1"})
    for a, b in zip(ref[:4], got[:4]):
        assert torch.equal(a, b)
    assert ref[4] == (0, 0)
    assert got[4][0] == steps and got[4][1] == steps          # one watchdog event and one env finished by the completion launch per step
    clean = run({})
    assert clean[4] == (0, 0)
    for a, b in zip(ref[:4], clean[:4]):
        assert torch.equal(a, b)
    # the product library does not act on the test variable
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.delenv("BP_PROF")
    monkeypatch.delenv("BP_PROF_LIB")
    prod = run({"BP_SCHED_DEBUG_DROP": "1"})
    assert prod[4] == (0, 0)
    for a, b in zip(ref[:4], prod[:4]):
        assert torch.equal(a, b)
    monkeypatch.setattr(_lib, "_lib", None)   # whatever was loaded last is dropped: the next test loads the product library afresh


def test_gym_adapter_surface_and_metric_plumbing():
    import benchpush_amd
    from benchpush_amd.envs.ship_ice import default_trials
    from benchpush_amd.metrics import ShipIceMetric
    trials = default_trials(0.1, 2, base_seed=1)
    env = benchpush_amd.make("ship-ice-v0", cfg={"concentration": 0.1}, trials=trials)
    u = env.unwrapped
    assert u.observation_space.shape == (4, 150, 150) and u.action_space.shape == ()
    assert u.goal == (0, 9) and abs(u.max_yaw_rate_step - (math.pi / 2) / 7) < 1e-15 and u.cfg.ship.mass == 1
    metric = ShipIceMetric("test", ship_mass=u.cfg.ship.mass, goal=u.goal)
    obs, info = u.reset()
    assert obs.dtype == np.uint8 and obs.shape == (4, 150, 150)
    assert set(info) == {"state", "total_work", "obs"} and len(info["obs"]) == len(trials[0]["obstacles"])
    metric.reset(info)
    for t in range(40):
        obs, r, done, trunc, info = u.step(np.float32(0.0))
        assert set(info) == {"state", "total_work", "collision reward", "scaled collision reward", "dist reward",
                             "trial_success", "obs"}
        metric.update(info, r, done or trunc)
        if done:
            break
    assert done and len(metric.efficiency_scores) == 1 and 0 < metric.effort_scores[0] <= 1
    obs2, info2 = u.reset()   # second episode -> next trial
    assert len(info2["obs"]) == len(trials[1]["obstacles"])
    env.close()


def test_cabi_call_order_errors():
    from benchpush_amd import _lib
    from benchpush_amd.config import default_cfg, ship_ice_physics_params
    cfg = default_cfg("ship_ice")
    L = _lib.load()
    bc = _lib.make_config(ship_ice_physics_params(cfg), cfg.ship.vertices, cfg.ship.head, cfg.ship.tail)
    h = C.c_void_p()
    assert L.bp_create(C.byref(bc), 2, 0, 0, C.byref(h)) == 0
    a = torch.zeros(2, dtype=torch.float64, device="cuda:0")
    assert L.bp_step(h, C.c_void_p(a.data_ptr()), None, None, None, None, None, None) == -5   # BP_ESTATE
    assert L.bp_reset(h, None, None, None, None) == -5
    assert b"bp_load_scenarios" in L.bp_last_error(h)
    for dp in (1.5, -0.1, float("nan")):   # pow(space.damping, dt) of a damping in [0, 1]; anything else is refused (0.5 is served: test_parity_nonzero_damping)
        bad = _lib.make_config(dict(ship_ice_physics_params(cfg), damping_pow=dp), cfg.ship.vertices, cfg.ship.head, cfg.ship.tail)
        h2 = C.c_void_p()
        assert L.bp_create(C.byref(bad), 2, 0, 0, C.byref(h2)) == -1                          # BP_EINVAL
    assert L.bp_destroy(h) == 0


def test_vec_env_protocol_autoreset_and_timelimit():
    """SB3-shaped VecEnv over the batched env: shapes, auto-reset with terminal_observation, TimeLimit truncation."""
    from benchpush_amd.envs.ship_ice import default_trials
    from benchpush_amd.envs.vec_env import make_ship_ice_vec_env
    trials = default_trials(0.1, 3, base_seed=2)
    venv = make_ship_ice_vec_env(6, cfg={"concentration": 0.1}, trials=trials)
    venv.max_episode_steps = 20
    obs = venv.reset()
    assert obs.shape == (6, 4, 150, 150) and obs.dtype == np.uint8 and venv.num_envs == 6
    first = obs.copy()
    saw_trunc = saw_term = False
    for t in range(45):
        a = np.zeros(6, np.float32)
        a[0] = 1.0                                   # env 0 steers into the channel boundary: terminated, not truncated
        obs, rew, done, infos = venv.step(a)
        assert obs.shape == (6, 4, 150, 150) and rew.shape == (6,) and done.shape == (6,) and len(infos) == 6
        for e in range(6):
            if done[e]:
                assert infos[e]["terminal_observation"].shape == (4, 150, 150)
                saw_trunc |= infos[e]["TimeLimit.truncated"]
                saw_term |= not infos[e]["TimeLimit.truncated"]
            else:
                assert "terminal_observation" not in infos[e]
        if t == 19:
            assert done[1:].all()                    # TimeLimit after 20 steps for the envs still running
    assert saw_trunc and saw_term
    assert set(infos[0]) >= {"x", "y", "theta", "total_work", "trial_success"}
    venv.close()


def test_vec_env_device_mode_matches_numpy_mode_and_keeps_terminal_observations():
    """`to_numpy=False` (no host synchronisation per step: device done mask into the masked reset, terminal rows saved by bp_copy_rows_masked, lazy infos)
    returns exactly what the numpy mode returns, and a finished env's `terminal_observation` is the raster of its last state -- the row the raw env
    produced before the auto-reset overwrote it."""
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
    from benchpush_amd.envs.vec_env import LazyInfos, make_ship_ice_vec_env
    trials = default_trials(0.1, 3, base_seed=2)
    vd = make_ship_ice_vec_env(6, cfg={"concentration": 0.1}, trials=trials, to_numpy=False)
    vn = make_ship_ice_vec_env(6, cfg={"concentration": 0.1}, trials=trials)
    raw = BatchedShipIceEnv(6, cfg={"concentration": 0.1}, trials=trials)
    vd.max_episode_steps = vn.max_episode_steps = 12
    od, on = vd.reset(), vn.reset()
    raw.reset()
    rsteps = np.zeros(6, int)
    assert torch.is_tensor(od) and od.is_cuda and np.array_equal(od.cpu().numpy(), on)
    ndone = 0
    for t in range(30):
        a = np.zeros(6, np.float32)
        a[0] = 1.0                                   # env 0 leaves the channel: terminated
        od, rd, dd, idv = vd.step(torch.from_numpy(a).cuda())
        on, rn, dn, inn = vn.step(a)
        ro, rr, rt, _, ri = raw.step(torch.from_numpy(a))
        ro = ro.cpu().numpy().copy()
        rsteps += 1
        assert torch.is_tensor(dd) and dd.is_cuda and dd.dtype == torch.bool and rd.is_cuda and isinstance(idv, LazyInfos) and len(idv) == 6
        assert np.array_equal(od.cpu().numpy(), on) and np.array_equal(rd.cpu().numpy(), rn) and np.array_equal(dd.cpu().numpy(), dn)
        rdone = rt.cpu().numpy().astype(bool) | (rsteps >= 12)
        assert np.array_equal(dn, rdone)
        for e in range(6):
            assert idv[e]["x"] == inn[e]["x"] == float(ri[e, 0]) and idv[-6 + e] is idv[e]
            if dn[e]:
                ndone += 1
                assert np.array_equal(inn[e]["terminal_observation"], ro[e])                       # the last observation of the finished episode
                assert torch.is_tensor(idv[e]["terminal_observation"]) and np.array_equal(idv[e]["terminal_observation"].cpu().numpy(), ro[e])
                assert idv[e]["TimeLimit.truncated"] == inn[e]["TimeLimit.truncated"] == (not bool(rt[e]))
            else:
                assert "terminal_observation" not in idv[e] and np.array_equal(on[e], ro[e])
        assert list(idv.done_indices()) == list(np.nonzero(dn)[0]) and [d["y"] for d in inn[1:3]] == [inn[1]["y"], inn[2]["y"]]
        raw.reset(torch.from_numpy(rdone))
        rsteps[rdone] = 0
    assert ndone >= 8
    vd.close(); vn.close(); raw.close()


def test_vec_env_previous_observation_survives_the_next_step():
    """SB3's collect_rollouts takes `new_obs = env.step()` first and stores `self._last_obs` -- the batch of the PREVIOUS step -- afterwards
    (DummyVecEnv deep-copies its buffer).  The numpy adapter must therefore never rewrite the batch of step t during step t + 1; infos kept for one
    more step must not change either (ADVICE r5, high + low)."""
    from benchpush_amd.envs.ship_ice import default_trials
    from benchpush_amd.envs.vec_env import make_ship_ice_vec_env
    trials = default_trials(0.1, 3, base_seed=2)
    for bufs in (2, 0, 3):
        venv = make_ship_ice_vec_env(6, cfg={"concentration": 0.1}, trials=trials)
        venv.obs_buffers = bufs
        venv.max_episode_steps = 5
        last = venv.reset()
        keep = last.copy()
        prev_infos = prev_snapshot = None
        for t in range(14):
            a = np.zeros(6, np.float32)
            a[0] = 1.0
            obs, rew, done, infos = venv.step(a)
            assert np.array_equal(last, keep), "the previous step's observation batch was rewritten by this step"
            assert not np.shares_memory(obs, last)
            if prev_infos is not None:                                    # the infos of step t - 1, read again after step t
                now = [(prev_infos[e]["x"], prev_infos[e]["y"], prev_infos[e].get("TimeLimit.truncated")) for e in range(6)]
                assert now == prev_snapshot[0]
                for e, to in prev_snapshot[1].items():
                    assert np.array_equal(prev_infos[e]["terminal_observation"], to)
            snap = [(infos[e]["x"], infos[e]["y"], infos[e].get("TimeLimit.truncated")) for e in range(6)]
            prev_infos, prev_snapshot = infos, (snap, {e: infos[e]["terminal_observation"].copy() for e in range(6) if done[e]})
            rew0, done0 = rew.copy(), done.copy()
            last, keep = obs, obs.copy()
        obs, rew, done, infos = venv.step(np.zeros(6, np.float32))
        assert np.array_equal(rew0, rew0.copy()) and rew is not rew0 and not np.shares_memory(done, done0)
        venv.close()


def test_log_obs_dumps_the_observation_channels(tmp_path):
    """cfg.log_obs on the single-env adapters: the files the reference's log_observation writes (names, directory per episode, vertical flip) hold the
    channels of the observation step() returned."""
    import benchpush_amd
    from benchpush_amd.envs.ship_ice import default_trials
    from benchpush_amd.obs_log import read_gray_png
    trials = default_trials(0.1, 2, base_seed=2)
    out = str(tmp_path / "logs")
    env = benchpush_amd.make("ship-ice-v0", cfg={"concentration": 0.1, "log_obs": True, "output_dir": out}, trials=trials).unwrapped
    env.reset()
    obs, _, _, _, _ = env.step(0.3)
    obs2, _, _, _, _ = env.step(-0.2)
    for t, o in ((1, obs), (2, obs2)):
        for name, c in (("footprint", 0), ("edt", 1), ("orientation", 2), ("con", 3)):
            assert np.array_equal(read_gray_png(os.path.join(out, "t0", "%d_%s.png" % (t, name))), o[c][::-1])
    env.reset()
    env.step(0.0)
    assert os.path.exists(os.path.join(out, "t1", "1_con.png")) and not os.path.exists(os.path.join(out, "t1", "2_con.png"))
    env.close()
    genv = benchpush_amd.make("ship-ice-v0", cfg={"concentration": 0.1, "log_obs": True, "output_dir": out + "_g", "egocentric_obs": False}, trials=trials).unwrapped
    genv.reset()
    gobs, _, _, _, _ = genv.step(0.1)
    assert np.array_equal(read_gray_png(os.path.join(out + "_g", "t0", "1_con.png")), gobs[0][::-1])
    assert np.array_equal(read_gray_png(os.path.join(out + "_g", "t0", "1_footprint.png")), gobs[1][::-1])
    genv.close()
    menv = benchpush_amd.make("maze-NAMO-v0", cfg={"log_obs": True, "output_dir": out + "_m"}).unwrapped
    menv.reset()
    mobs, _, _, _, _ = menv.step(0.2)
    for name, c in (("footprint", 0), ("movable_obs", 1), ("fixed_obs", 2), ("local_distance_map", 3)):
        assert np.array_equal(read_gray_png(os.path.join(out + "_m", "t0", "1_%s.png" % name)), mobs[c][::-1])
    g = read_gray_png(os.path.join(out + "_m", "t0", "1_distance_map.png"))
    assert g.shape == menv._goal_dt.shape and g.max() == 255
    menv.close()
    with pytest.raises(NotImplementedError):
        benchpush_amd.make("box-delivery-v0", cfg={"render": {"log_obs": True}})


def test_global_planner_observation_matches_oracle():
    """cfg.egocentric_obs: false -> uint8 [2, 200, 60] (5x5 block-mean occupancy + footprint), byte for byte."""
    import benchpush_amd
    from benchpush_amd.envs.ship_ice import default_trials
    trials = default_trials(0.3, 2, base_seed=6)
    env = _mk(2, 0.3, trials)
    env.reset()
    orcs = _oracles(env, 2)
    for e, o in enumerate(orcs):
        o.reset(trials[e % 2], observe=False)
    for t in range(6):
        a = np.array([0.3, -0.5], np.float32).astype(np.float64)
        env.step(torch.from_numpy(a))
        g = env.observe_global().cpu().numpy()
        assert g.shape == (2, 2, 200, 60)
        for e, o in enumerate(orcs):
            o.step(float(a[e]), observe=False)
            og = o.observe_global()
            assert np.array_equal(g[e], og), (t, e)
    assert 0.2 < g[0, 0].mean() / 255 < 0.4 and (g[0, 1] == 255).sum() > 5
    single = benchpush_amd.make("ship-ice-v0", cfg={"concentration": 0.3, "egocentric_obs": False}, trials=trials).unwrapped
    obs, info = single.reset()
    assert obs.shape == (2, 200, 60) and single.observation_space.shape == (2, 200, 60)
    single.close()
