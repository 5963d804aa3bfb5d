"""Distribution of per-env wave cycles of k_physics_step vs the launch time: python tools/env_cost_dist.py [E] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
CONC = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
trials = default_trials(CONC, 100, base_seed=0)
env = BatchedShipIceEnv(E, cfg={"concentration": CONC}, trials=trials)
env.reset()
g = torch.Generator(device=env.device); g.manual_seed(1234)
for t in range(STEPS):
    a = (torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1).float().double()
    env.enable_timing(True)
    _, _, term, _, _ = env.step(a)
    p_ms, r_ms, n = env.kernel_time_ms()
    c = env.step_cycles().astype(np.float64)
    if t % 4 == 3:
        q = np.percentile(c, [50, 90, 99, 100]) / 1e6
        print("step %2d physics %.2f ms | env Mcycles mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | sum/2048 slots = %.1f Mcyc, at 2.4GHz: max %.1f ms, packed %.1f ms"
              % (t, p_ms, c.mean() / 1e6, q[0], q[1], q[2], q[3], c.sum() / 2048 / 1e6, q[3] / 2.4e3 * 1e3 / 1e3, c.sum() / 2048 / 2.4e6))
    env.reset(term)
