#!/usr/bin/env python3
"""Phase breakdown of the paired sub-step (bp_physics_pair.hpp) from the -DBP_PAIR_PROF diagnostic build:
    tools/build_variant.sh pairprof "-DBP_PAIR_PROF=1" && BP_PROF=1 BP_PROF_LIB=benchpush_amd/libbenchpush_hip_pairprof.so BP_PAIR=1 python tools/prof_pair.py [E] [steps]
Cycle stamps are taken by lane 0 for the wave (both halves); fixed pairs (BP_PAIR=1) keep both halves running for the whole step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 28
trials = default_trials(0.3, 100, base_seed=0)
env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials)
print("pair mode", env.L.bp_pair_mode(env.h))
env.reset()
prof = torch.zeros((E, 64), dtype=torch.int64, device=env.device)
env.L.bp_debug_prof(env.h, prof.data_ptr())
g = torch.Generator(device=env.device); g.manual_seed(1234)
names = ["integrate", "cand+cached", "bounds+search", "manifold", "deliver", "filter+prestep+warm", "colouring", "velint+warmstart", "solver", "post+rules", "mvlist"]
for t in range(STEPS):
    a = (torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1).float().double()
    prof.zero_()
    _, _, term, _, _ = env.step(a)
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.float64)
    if t >= STEPS - 3:
        rows = p[p[:, 12] > 0]
        tot = rows[:, 11]
        nsub = rows[:, 12]
        print("step %d: %d waves, wave cycles mean %.0f max %.0f; sub-steps per wave %.1f" % (t, len(rows), tot.mean(), tot.max(), nsub.mean()))
        m = rows.sum(0)
        print("   share of the phases: " + " ".join("%s=%.1f%%" % (n, 100 * m[i] / m[:11].sum()) for i, n in enumerate(names)))
        print("   cycles per wave sub-step: " + " ".join("%s=%.0f" % (n, m[i] / m[12]) for i, n in enumerate(names)) + " | sum=%.0f" % (m[:11].sum() / m[12]))
        print("   per wave sub-step: candidate rounds %.2f, colour passes (slot loop) %.2f, single-colour iterations %.2f" % (m[13] / m[12], m[14] / m[12], m[15] / m[12]))
    env.reset(term)
