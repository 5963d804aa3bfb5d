"""prof_phases.py for a library of the round-2 layout (24 counters per env, overwritten per run): BP_SCHED=0 BP_PROF=1 BP_PROF_LIB=... python tools/prof_phases24.py"""
import os, sys
os.environ["BP_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
E = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 14
trials = default_trials(0.3, 100, base_seed=0)
env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials)
env.reset()
prof = torch.zeros((E, 24), dtype=torch.int64, device=env.device)
env.L.bp_debug_prof(env.h, prof.data_ptr())
g = torch.Generator(device=env.device); g.manual_seed(1234)
names = ["integrate", "refresh", "cand+hint", "face_seps", "deliver", "filter", "prestep+warmset", "velint+warm", "solver", "post+mvlist", "manifolds"]
for t in range(STEPS):
    a = (torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1).float().double()
    _, _, term, _, _ = env.step(a)
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.float64)
    if t >= STEPS - 2:
        tot = p[:, 23]; worst = int(np.argmax(tot))
        for who, row in (("mean", p.mean(0)), ("worst", p[worst])):
            print("step %d %s cycles per sub-step: " % (t, who) + " ".join("%s=%.0f" % (n, row[i] / 400) for i, n in enumerate(names)) + " total=%.0f" % (row[23] / 400))
    env.reset(term)
