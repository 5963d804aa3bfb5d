"""Two environments per wavefront (benchpush_amd/csrc/bp_physics_pair.hpp), -m gpu: the paired sub-step -- lanes 0..31 one env, lanes 32..63 another, one
instruction stream -- against the CPU oracle and against the one-env-per-wavefront kernel, bit for bit.

BP_PAIR=1 runs every env step through k_physics_step_pair (fixed pairs (2b, 2b + 1) of the dispatch order for the whole step, no env ever leaves its
pair), BP_PAIR=2 is the scheduler-integrated product path (heavy envs alone, light envs in pairs, envs that outgrow the half-wave parked and resumed alone)."""
import numpy as np
import pytest
import torch

from test_gpu_parity import _run_parity

pytestmark = pytest.mark.gpu


def test_fixed_pairs_match_the_oracle_30pct(monkeypatch):
    """8 envs = 4 paired waves, 40 steps with contacts, auto-resets and later episodes: body state (==), observations, info, rewards, flags."""
    monkeypatch.setenv("BP_PAIR", "1")
    assert _run_parity(E=8, conc=0.3, T=3, steps=40, seed=0) > 1000


def test_fixed_pairs_match_the_oracle_50pct_and_ragged_last_wave(monkeypatch):
    """50 % concentration (more arbiters, colours and moving bodies per half) and an odd env count: the last wave carries one env and an idle half."""
    monkeypatch.setenv("BP_PAIR", "1")
    assert _run_parity(E=7, conc=0.5, T=2, steps=10, seed=21) > 100   # (beyond step 11 env 3 outgrows the 32 arbiter lanes of a half: fixed pairs cannot leave, the scheduler path can)
    assert _run_parity(E=5, conc=0.1, T=2, steps=20, seed=3) >= 0


def test_fixed_pairs_boundary_and_yaw_edges(monkeypatch):
    """hard-over actions: yaw clamp at 0 / pi, channel boundary, termination inside a pair while the mate carries on."""
    monkeypatch.setenv("BP_PAIR", "1")
    _run_parity(E=6, conc=0.2, T=2, steps=45, seed=8, action_fn=lambda e, t: (1.0, -1.0, 0.0, 0.7, -0.7, 0.2)[e])


def _run_batch(E, steps, conc, env_vars, monkeypatch, trials):
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv
    for k in ("BP_PAIR", "BP_SCHED", "BP_PP_ACT", "BP_PP_WORK", "BP_PAIR_SOLO", "BP_PP_KEYS", "BP_PP_SLOTS", "BP_PP_MV", "BP_PAIR_RESIDENT"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env_vars.items():
        monkeypatch.setenv(k, v)
    g = torch.Generator(device="cuda:0")
    g.manual_seed(33)
    acts = (torch.rand((steps, E), generator=g, device="cuda:0", dtype=torch.float64) * 2 - 1).float().double()
    env = BatchedShipIceEnv(E, cfg={"concentration": conc}, trials=trials, device="cuda:0")
    assert int(env.L.bp_pair_mode(env.h)) == int(env_vars.get("BP_PAIR", "0"))
    env.reset()
    rsum = torch.zeros(E, dtype=torch.float64, device="cuda:0")
    nterm = 0
    for t in range(steps):
        _, rew, term, _, _ = env.step(acts[t])
        rsum += rew
        nterm += int(term.sum().item())
        env.reset(term)
    errs = None
    try:
        env.check_errors()
    except Exception as e:  # noqa: BLE001
        errs = str(e)
    out = (env.body_state().clone(), rsum, env.obs.clone(), env.info.clone(), env.episode_metrics()[0].clone(), nterm, errs)
    env.close()
    return out


def test_fixed_pairs_equal_the_solo_kernel_on_512_envs(monkeypatch):
    """512 envs x 45 steps with auto-reset (every episode phase, incl. the heavy middle): fixed pairs against one wavefront per env, torch.equal on body
    state, summed rewards, observations, info and episode metrics; no capacity flag in either."""
    from benchpush_amd.envs.ship_ice import default_trials
    trials = default_trials(0.3, 24, base_seed=4)
    ref = _run_batch(512, 45, 0.3, {"BP_SCHED": "0"}, monkeypatch, trials)
    got = _run_batch(512, 45, 0.3, {"BP_PAIR": "1"}, monkeypatch, trials)
    assert ref[6] is None and got[6] is None, (ref[6], got[6])
    assert ref[5] == got[5] and ref[5] > 100
    for a, b in zip(ref[:5], got[:5]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("resident", ["1", "0"], ids=["resident_kernel_with_function_bodies", "dispatcher_driven_pair_of_kernels"])
def test_pairs_inside_the_scheduler_match_the_oracle(monkeypatch, resident):
    """BP_PAIR=2, the product path: paired first tasks, envs leaving their pair at a sub-step boundary (tight limits force it within every step), the
    heavier one carrying on alone in the same wavefront, the other resumed from the queues by another workgroup -- bit-identical to the oracle.  Both launch
    forms: k_physics_step_schedr (default: resident workgroups, the two step bodies as functions) and k_physics_step_sched + k_physics_step_schedp (BP_PAIR_RESIDENT=0)."""
    monkeypatch.setenv("BP_PAIR_RESIDENT", resident)
    monkeypatch.setenv("BP_PAIR", "2")
    monkeypatch.setenv("BP_PAIR_SOLO", "2")
    assert _run_parity(E=9, conc=0.3, T=3, steps=30, seed=5) > 500
    monkeypatch.setenv("BP_PP_ACT", "3")        # leave as soon as four arbiters are active: every pair splits, at arbitrary sub-steps
    monkeypatch.setenv("BP_PP_WORK", "2")
    monkeypatch.setenv("BP_PAIR_SOLO", "0")
    assert _run_parity(E=8, conc=0.3, T=3, steps=25, seed=0) > 500
    monkeypatch.setenv("BP_PP_ACT", "12")
    monkeypatch.setenv("BP_PP_WORK", "16")
    assert _run_parity(E=7, conc=0.5, T=2, steps=14, seed=21) > 100   # 50 %: the env that outgrows the half-wave at step 11 leaves its pair instead of overflowing


def test_pairs_inside_the_scheduler_equal_the_solo_kernel_at_full_size(monkeypatch):
    """4096 envs x 40 steps with auto-reset: the product path (BP_PAIR=2) and the same with limits that split most pairs against one wavefront per env
    (BP_SCHED=0): body state, summed rewards, observations, info, episode metrics torch.equal; no capacity flag, no scheduler warning."""
    from benchpush_amd.envs.ship_ice import default_trials
    trials = default_trials(0.3, 48, base_seed=1)
    ref = _run_batch(4096, 40, 0.3, {"BP_SCHED": "0"}, monkeypatch, trials)
    assert ref[6] is None and ref[5] > 500
    for variant in ({"BP_PAIR": "2"}, {"BP_PAIR": "2", "BP_PP_ACT": "6", "BP_PP_WORK": "6", "BP_PAIR_SOLO": "1000"}, {"BP_PAIR": "2", "BP_PAIR_RESIDENT": "0"}):
        got = _run_batch(4096, 40, 0.3, variant, monkeypatch, trials)
        assert got[6] is None, got[6]
        assert got[5] == ref[5]
        for a, b in zip(ref[:5], got[:5]):
            assert torch.equal(a, b), variant


def test_default_from_5120_envs_is_paired_and_equals_the_solo_kernel(monkeypatch):
    """Handles of 5120 envs and more pair by default (from 7168 up every first task is a pair and an env leaves at 20 active arbiters / 40 work units per sub-step; 5120 ... 7167:
    the tight limits, the heaviest eighth starts alone; on resident wavefronts, the two step bodies as functions of one kernel); below, the lean one-env-per-wavefront scheduler kernel stays, on resident wavefronts.  6144 envs x 14 steps with auto-reset
    against BP_PAIR=0: torch.equal."""
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
    trials = default_trials(0.3, 32, base_seed=9)
    for k in ("BP_PAIR", "BP_SCHED"):
        monkeypatch.delenv(k, raising=False)
    small = BatchedShipIceEnv(64, cfg={"concentration": 0.3}, trials=trials, device="cuda:0")
    assert int(small.L.bp_pair_mode(small.h)) == 0
    small.close()
    ref = _run_batch(6144, 14, 0.3, {"BP_PAIR": "0"}, monkeypatch, trials)
    got = _run_batch(6144, 14, 0.3, {"BP_PAIR": "2"}, monkeypatch, trials)      # = the default at this size (asserted below)
    assert ref[6] is None and got[6] is None, (ref[6], got[6])
    assert ref[5] == got[5]
    for a, b in zip(ref[:5], got[:5]):
        assert torch.equal(a, b)
    for k in ("BP_PAIR", "BP_SCHED"):
        monkeypatch.delenv(k, raising=False)
    big = BatchedShipIceEnv(6144, cfg={"concentration": 0.3}, trials=trials, device="cuda:0")
    ps = big.pair_stats()
    assert ps["mode"] == 2 and ps["solo_first"] == 768 and ps["max_warm_x_colours"] == 9
    assert int(big.L.bp_sched_resident(big.h)) > 0           # pairing launches run on resident wavefronts too (k_physics_step_schedr)
    big.close()
    bigger = BatchedShipIceEnv(7168, cfg={"concentration": 0.3}, trials=trials, device="cuda:0")
    ps = bigger.pair_stats()
    assert ps["mode"] == 2 and ps["solo_first"] == 0 and ps["max_warm_x_colours"] == 40
    bigger.close()
    mid = BatchedShipIceEnv(5120, cfg={"concentration": 0.3}, trials=trials, device="cuda:0")
    ps = mid.pair_stats()
    assert ps["mode"] == 2 and ps["solo_first"] == 640 and ps["max_warm_x_colours"] == 9
    mid.close()
    below = BatchedShipIceEnv(4096, cfg={"concentration": 0.3}, trials=trials, device="cuda:0")
    assert int(below.L.bp_pair_mode(below.h)) == 0 and int(below.L.bp_sched_resident(below.h)) > 0
    below.close()
