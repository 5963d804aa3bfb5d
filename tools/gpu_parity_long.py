"""Longer GPU <-> oracle sweep than the test-suite runs (body state, reward, termination every step, an observation every 10 steps, resets included).
    python tools/gpu_parity_long.py [E] [steps] [concentration] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
from oracle.oracle import OracleShipIce

E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 150
CONC = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
SEED = int(sys.argv[4]) if len(sys.argv) > 4 else 123
T = 8
trials = default_trials(CONC, T, base_seed=SEED)
env = BatchedShipIceEnv(E, cfg={"concentration": CONC}, trials=trials, device="cuda:0")
c = env.cfg
orcs = [OracleShipIce(env.params, c.ship.vertices, c.ship.head, c.ship.tail) for _ in range(E)]
obs, _ = env.reset()
eps = [0] * E
for e in range(E):
    oo, _ = orcs[e].reset(trials[e % T])
    assert np.array_equal(obs[e].cpu().numpy(), oo)
rng = np.random.default_rng(SEED)
age = np.zeros(E, int)
t0 = time.time()
nreset = 0
for t in range(STEPS):
    a = rng.uniform(-1, 1, E) * (0.35 if t % 40 < 30 else 1.0)
    a[rng.random(E) < 0.05] = 0.0
    obs, rew, term, _, info = env.step(torch.from_numpy(a))
    bs = env.body_state().cpu().numpy()
    tm = term.cpu().numpy().astype(bool)
    ob = obs.cpu().numpy() if t % 10 == 0 else None
    for e in range(E):
        oo, orr, ot, _ = orcs[e].step(float(a[e]), observe=ob is not None)
        nb = len(orcs[e].bodies())
        assert np.array_equal(bs[e, :nb], orcs[e].bodies()), ("bodies", t, e)
        assert float(rew[e]) == orr and bool(tm[e]) == ot, ("reward / termination", t, e)
        if ob is not None:
            assert np.array_equal(ob[e], oo), ("observation", t, e)
    age += 1
    m = tm | (age >= 300)
    if m.any():
        obs, _ = env.reset(torch.from_numpy(m.astype(np.uint8)))
        for e in range(E):
            if m[e]:
                eps[e] += 1; age[e] = 0; nreset += 1
                oo, _ = orcs[e].reset(trials[(e + eps[e]) % T])
                assert np.array_equal(obs[e].cpu().numpy(), oo), ("reset observation", t, e)
    if t % 25 == 24:
        print("step %d ok (%d resets, %.0f s)" % (t + 1, nreset, time.time() - t0), flush=True)
env.check_errors()
print("bit-identical: %d envs x %d steps at %.0f %%, %d resets" % (E, STEPS, CONC * 100, nreset))
