"""How well does last step's per-env cost predict this step's heaviest envs?  python tools/cost_predictability.py [E] [warm] [steps]
(the dispatch order of k_physics_step and every solo / mixed launch rely on it)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
WARM = int(sys.argv[2]) if len(sys.argv) > 2 else 40
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 12
trials = default_trials(0.3, 100, base_seed=0)
env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials)
env.reset()
g = torch.Generator(device=env.device); g.manual_seed(1234)
prev = None
for t in range(WARM + STEPS):
    a = (torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1).float().double()
    env.enable_timing(True)
    _, _, term, _, _ = env.step(a)
    p_ms, _, _ = env.kernel_time_ms()
    c = env.step_cycles().astype(np.float64)
    if prev is not None and t >= WARM:
        order_prev = np.argsort(-prev)                 # dispatch order used for this step
        rank_prev = np.empty(E, int); rank_prev[order_prev] = np.arange(E)
        top = np.argsort(-c)
        line = "step %3d physics %.2f ms max %.1f M p99 %.1f M | rank (in last step's order) of this step's heaviest 1/2/4/8: %s" % (
            t, p_ms, c.max() / 1e6, np.percentile(c, 99) / 1e6, " ".join(str(rank_prev[top[k]]) for k in (0, 1, 3, 7)))
        for n in (64, 128, 256, 512):
            line += " | top-64 inside predicted top-%d: %.0f%%" % (n, 100 * np.mean(rank_prev[top[:64]] < n))
        # the heaviest env that the predicted top-n misses: what a solo set of n envs leaves as the tail
        for n in (128, 512):
            miss = c[rank_prev >= n].max()
            line += " | heaviest outside predicted top-%d: %.1f M" % (n, miss / 1e6)
        print(line)
    prev = c.copy()
    # envs that reset this step start a new episode: their cost next step is a fresh episode's
    env.reset(term)
