"""Observation parity sweep: python tools/gpu_parity_obs.py [E] [steps] [conc]  -- GPU k_observe vs the oracle's raster, byte for byte,
on E envs x steps env.step() with random actions (oracle stepped alongside, multiprocessing over envs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multiprocessing import Pool
from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
CONC = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
T = 16
trials = default_trials(CONC, T, base_seed=5)
rng = np.random.default_rng(9)
acts = rng.uniform(-1, 1, (STEPS, E)).astype(np.float32).astype(np.float64)


def oracle_run(e):
    from benchpush_amd.config import default_cfg, ship_ice_physics_params
    from oracle.oracle import OracleShipIce
    cfg = default_cfg("ship_ice"); cfg.concentration = CONC
    o = OracleShipIce(ship_ice_physics_params(cfg), cfg.ship.vertices, cfg.ship.head, cfg.ship.tail)
    out = [o.reset(trials[e % T])[0]]
    for t in range(STEPS):
        ob, r, term, _ = o.step(float(acts[t, e]))
        out.append(ob)
        if term:
            break
    return np.stack(out)


if __name__ == "__main__":
    with Pool(min(16, os.cpu_count())) as p:
        ref = p.map(oracle_run, range(E))
    env = BatchedShipIceEnv(E, cfg={"concentration": CONC}, trials=trials)
    obs, _ = env.reset()
    alive = np.ones(E, bool)
    bad = 0; npx = 0; nocc = 0
    g = obs.cpu().numpy()
    for e in range(E):
        bad += int((g[e] != ref[e][0]).sum()); npx += g[e].size
    for t in range(STEPS):
        obs, rew, term, _, _ = env.step(torch.from_numpy(acts[t]))
        g = obs.cpu().numpy()
        for e in range(E):
            if alive[e] and t + 1 < len(ref[e]):
                d = int((g[e] != ref[e][t + 1]).sum())
                if d:
                    print("MISMATCH env %d step %d: %d bytes, channels %s" % (e, t, d, [(g[e][c] != ref[e][t + 1][c]).sum() for c in range(4)]))
                bad += d; npx += g[e].size; nocc += int((g[e][3] == 255).sum())
        alive &= ~term.cpu().numpy().astype(bool)
    env.check_errors()
    print("observation parity: %d mismatching bytes of %d compared (%d occupied pixels)" % (bad, npx, nocc))
    sys.exit(1 if bad else 0)
