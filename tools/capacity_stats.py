"""Capacity demand of k_physics_step (BP_PROF build): max arbiter slots / active / warm arbiters / velocity slots / moving bodies
per env over an episode: BP_PROF=1 python tools/capacity_stats.py [E] [steps] [concentration]"""
import os
import sys

os.environ["BP_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials

E = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 80
CONC = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
trials = default_trials(CONC, 100, base_seed=0)
env = BatchedShipIceEnv(E, cfg={"concentration": CONC}, trials=trials)
env.reset()
prof = torch.zeros((E, 24), dtype=torch.int64, device=env.device)
env.L.bp_debug_prof(env.h, prof.data_ptr())
g = torch.Generator(device=env.device); g.manual_seed(1234)
names = {11: "arbiter slots used", 12: "active arbiters", 13: "warm arbiters", 14: "velocity slots", 15: "moving bodies"}
mx = np.zeros((24,), dtype=np.float64)
for t in range(STEPS):
    a = (torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1).float().double()
    _, _, term, _, _ = env.step(a)
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.float64)
    mx = np.maximum(mx, p.max(0))
    if t % 10 == 9 or t == STEPS - 1:
        print("step %d conc %.2f: " % (t, CONC) + "; ".join("%s p50 %.0f p99 %.0f max %.0f (run max %.0f)" % (
            n, np.percentile(p[:, k], 50), np.percentile(p[:, k], 99), p[:, k].max(), mx[k]) for k, n in names.items()), flush=True)
    env.reset(term)
env.check_errors()
