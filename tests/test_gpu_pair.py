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
    assert _run_parity(E=7, conc=0.5, T=2, steps=10, seed=21) > 100   # (step 11 of env 3 outgrows the half-wave: fixed pairs cannot leave, the scheduler path can)
    assert _run_parity(E=5, conc=0.1, T=2, steps=20, seed=3) >= 0


def test_fixed_pairs_boundary_and_yaw_edges(monkeypatch):
    """hard-over actions: yaw clamp at 0 / pi, channel boundary, termination inside a pair while the mate carries on."""
    monkeypatch.setenv("BP_PAIR", "1")
    _run_parity(E=6, conc=0.2, T=2, steps=45, seed=8, action_fn=lambda e, t: (1.0, -1.0, 0.0, 0.7, -0.7, 0.2)[e])


def _run_batch(E, steps, conc, env_vars, monkeypatch, trials):
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv
    for k in ("BP_PAIR", "BP_SCHED"):
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
