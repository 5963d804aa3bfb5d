"""Differential fuzz of the three restatements of the sub-step: oracle/bp_oracle.c, the one-env-per-wavefront HIP sub-step
(csrc/bp_physics.hpp) and the two-envs-per-wavefront one (csrc/bp_physics_pair.hpp), on 4 000 random small scenes (tests/fuzz_scenes.py: 0-12 bodies,
constructed parallel-edge and corner-to-corner contacts, overlaps from the first sub-step, shape radii 0 and 0.02) x (1 settle sub-step + 3 env steps of
40 sub-steps).

Bar: `==` on binary64 body state (pose, velocities, bias velocities of every shape), rewards, termination flags and the info block (contact counters,
impulse / kinetic-energy sums) after every env step.  A fix that reaches only one or two of the three copies shows here as a mismatch on a named scene.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fuzz_scenes import fuzz_params, make_scenes  # noqa: E402

pytestmark = pytest.mark.gpu

SUBSTEPS, STEPS, PER_RADIUS = 40, 3, int(os.environ.get("BP_FUZZ_SCENES", "2000"))   # BP_FUZZ_SCENES=<n per radius> for a longer one-off sweep


def _oracle_run(params, cfg, scenes, actions):
    from oracle.oracle import OracleShipIce
    o = OracleShipIce(params, cfg.ship.vertices, cfg.ship.head, cfg.ship.tail)
    out = []
    for e, sc in enumerate(scenes):
        o.reset(sc, observe=False)
        rec = []
        for t in range(STEPS):
            _, r, term, info = o.step(float(actions[t, e]), observe=False)
            rec.append((o.bodies().copy(), r, term, np.array(list(info.values()))))
        out.append(rec)
    return out


def _gpu_run(monkeypatch, envvars, radius, scenes, actions):
    import benchpush_amd.envs.ship_ice as si
    for k in ("BP_PAIR", "BP_SCHED", "BP_SCHED_PERSIST"):
        monkeypatch.delenv(k, raising=False)
    for k, v in envvars.items():
        monkeypatch.setenv(k, v)
    real = si.ship_ice_physics_params
    monkeypatch.setattr(si, "ship_ice_physics_params", lambda cfg: fuzz_params(real(cfg), radius, SUBSTEPS))
    E = len(scenes)
    env = si.BatchedShipIceEnv(E, cfg={"concentration": 0.1}, trials=scenes, device="cuda:0")   # env e plays scene e (episode 0: trial (e + 0) % T)
    monkeypatch.setattr(si, "ship_ice_physics_params", real)
    assert env.params["settle_steps"] == 1 and env.params["steps"] == SUBSTEPS
    want_pair = int(envvars.get("BP_PAIR", "0"))
    assert int(env.L.bp_pair_mode(env.h)) == want_pair, "the fuzz must run the kernel it names"
    env.reset()
    rec = []
    for t in range(STEPS):
        _, rew, term, _, info = env.step(torch.from_numpy(actions[t]))
        rec.append((env.body_state().cpu().numpy().copy(), env.num_bodies().copy(), rew.cpu().numpy().copy(), term.cpu().numpy().copy(), info.cpu().numpy().copy()))
    env.check_errors()
    params = dict(env.params)
    cfg = env.cfg
    env.close()
    return rec, params, cfg


KERNELS = [("scheduled, resident (default)", {"BP_PAIR": "0"}),
           ("one wavefront per env, no scheduler", {"BP_PAIR": "0", "BP_SCHED": "0"}),
           ("fixed pairs (k_physics_step_pair)", {"BP_PAIR": "1"}),
           ("pairs inside the scheduler", {"BP_PAIR": "2"})]


@pytest.mark.parametrize("radius", [0.02, 0.0])
def test_differential_fuzz_oracle_vs_solo_vs_paired(monkeypatch, radius):
    scenes = make_scenes(PER_RADIUS, radius, base_seed=0 if radius else 100000)
    rng = np.random.default_rng(7 + int(radius * 1000))
    actions = rng.uniform(-1, 1, (STEPS, PER_RADIUS)).astype(np.float32).astype(np.float64)
    ref = None
    ncontact = 0
    for name, envvars in KERNELS:
        rec, params, cfg = _gpu_run(monkeypatch, envvars, radius, scenes, actions)
        if ref is None:
            assert params["poly_radius"] == radius
            ref = _oracle_run(params, cfg, scenes, actions)
        for t in range(STEPS):
            bs, nb, rew, term, info = rec[t]
            for e in range(PER_RADIUS):
                ob, orr, ot, oi = ref[e][t]
                assert nb[e] == len(ob), (name, "bodies", e)
                if not np.array_equal(bs[e, : nb[e]], ob):
                    bad = np.nonzero((bs[e, : nb[e]] != ob).any(axis=1))[0]
                    raise AssertionError("%s: scene %d (family %d, radius %g), step %d: body state differs from the oracle for shapes %s" % (name, e, e % 4, radius, t, bad.tolist()))
                assert rew[e] == orr and bool(term[e]) == ot, (name, "reward / termination", e, t)
                assert np.array_equal(info[e], oi), (name, "info", e, t)
                ncontact += int(oi[14])
    assert ncontact > 50 * PER_RADIUS   # the scenes do collide: ship x floe contact points over all scenes, steps and kernels


def test_fuzz_scenes_cover_the_constructed_families():
    """The generator itself (no GPU work beyond the marker): families 1-3 put exactly-parallel, nearly-parallel and corner-to-corner pairs into the batch."""
    sc = make_scenes(400, 0.02)
    nb = np.array([len(s["obstacles"]) for s in sc])
    assert nb.min() == 0 and nb.max() == 12
    fam1 = [s for i, s in enumerate(sc) if i % 4 == 1]
    ang = []
    for s in fam1:
        a, b = s["obstacles"][0]["vertices"], s["obstacles"][1]["vertices"]
        ea, eb = a[1] - a[0], b[1] - b[0]
        ang.append(abs(np.arctan2(ea[0] * eb[1] - ea[1] * eb[0], ea @ eb)))
    ang = np.array(ang)
    assert (ang < 1e-14).sum() >= 5 and ((ang > 1e-13) & (ang < 1e-8)).sum() >= 5 and (ang > 1e-5).sum() >= 5   # exactly parallel up to the rounding of the corner coordinates, 1e-12 / 1e-9, 1e-4


@pytest.mark.parametrize("version", [1, 2])
def test_differential_fuzz_maze_oracle_vs_hip(monkeypatch, version):
    """The same idea for maze-NAMO-v0 (substep<BP_ENV_MAZE>: 8-vertex loops, the 5-shape kinematic robot, wall segments of radius 0.5 as 2-vertex hulls, the
    (1,3) flag-only pairs): 1 500 layouts per maze version whose 20 boxes are dropped ANYWHERE -- on walls, on each other, on the robot -- with a random start
    pose, one settle sub-step, three env steps of 40 sub-steps; scheduled resident kernel and plain kernel against the oracle with `==` on every shape's state,
    rewards, flags (wall collisions) and info."""
    import benchpush_amd.envs.maze_namo as mz
    from benchpush_amd.config import default_cfg, maze_walls, merge_user_cfg
    from oracle.oracle import OracleMaze
    N, nbox = 1500, 20
    cfg0 = mz._maze_cfg({"num_obstacles": nbox, "maze_version": version})
    walls = maze_walls(cfg0)
    W, Lh = float(cfg0.env.width), float(cfg0.env.length)
    rng = np.random.default_rng(900 + version)
    layouts = []
    for i in range(N):
        start = np.array([rng.uniform(1.0, W - 1.0), rng.uniform(1.0, Lh - 1.0), rng.uniform(-math.pi, math.pi)])
        c = np.stack([rng.uniform(0.3, W - 0.3, nbox), rng.uniform(0.3, Lh - 0.3, nbox)], -1)
        k = int(rng.integers(0, 6))                      # a few boxes right at the robot, so that most layouts push something from the first sub-step
        c[:k] = start[:2] + rng.uniform(-1.2, 1.2, (k, 2))
        layouts.append({"centres": c, "walls": np.array(walls, np.float64), "start": start})
    actions = rng.uniform(-1, 1, (STEPS, N)).astype(np.float32).astype(np.float64)
    real = mz.maze_physics_params

    def params_of(cfg):
        p = dict(real(cfg))
        p.update(settle_steps=1, steps=SUBSTEPS, dt=float(p["dt"]) / int(p["steps"]) * SUBSTEPS)
        return p

    ref = None
    nwall = 0
    for name, envvars in (("scheduled, resident (default)", {}), ("one wavefront per env, no scheduler", {"BP_SCHED": "0"})):
        for k in ("BP_SCHED", "BP_SCHED_PERSIST"):
            monkeypatch.delenv(k, raising=False)
        for k, v in envvars.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setattr(mz, "maze_physics_params", params_of)
        env = mz.BatchedMazeEnv(N, cfg={"num_obstacles": nbox, "maze_version": version}, layouts=layouts, device="cuda:0")
        monkeypatch.setattr(mz, "maze_physics_params", real)
        assert env.params["settle_steps"] == 1 and env.params["steps"] == SUBSTEPS
        env.reset()
        rec = []
        for t in range(STEPS):
            _, rew, term, _, info = env.step(torch.from_numpy(actions[t]))
            rec.append((env.body_state().cpu().numpy().copy(), rew.cpu().numpy().copy(), term.cpu().numpy().copy(), info.cpu().numpy().copy()))
        env.check_errors()
        if ref is None:
            c = env.cfg
            o = OracleMaze(env.params, c.robot.vertices, c.robot.wheel_vertices, c.obstacle_size)
            ref = []
            for e in range(N):
                o.reset(layouts[e], observe=False)
                r_ = []
                for t in range(STEPS):
                    _, orr, ot, oi = o.step(float(actions[t, e]), observe=False)
                    r_.append((o.shape_states().copy(), orr, ot, np.array(list(oi.values()))))
                ref.append(r_)
        env.close()
        for t in range(STEPS):
            bs, rew, term, info = rec[t]
            for e in range(N):
                ss, orr, ot, oi = ref[e][t]
                if not np.array_equal(bs[e, : len(ss)], ss):
                    bad = np.nonzero((bs[e, : len(ss)] != ss).any(axis=1))[0]
                    raise AssertionError("%s: maze v%d layout %d, step %d: shape state differs from the oracle for shapes %s" % (name, version, e, t, bad.tolist()))
                assert rew[e] == orr and bool(term[e]) == ot, (name, "reward / termination", e, t)
                assert np.array_equal(info[e], oi), (name, "info", e, t)
                nwall += int(oi[10])
    assert nwall > 100     # layouts that start on or next to a wall do report the (1,3) wall collision
