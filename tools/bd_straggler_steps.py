"""Per env step of the box-delivery bench workload: the slowest env's wave cycles and sim steps, envs that ran into STEP_LIMIT, recurrences skipped.
    [BP_BD_CYCLE=0] python tools/bd_straggler_steps.py [E] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 13
env = BatchedBoxDeliveryEnv(E, cfg={"boxes": {"num_boxes_small": 12}}, num_trials=64)
env.reset()
g = torch.Generator(device=env.device)
g.manual_seed(1234)   # bench.py's action stream of rank 0
acts = (torch.rand((STEPS, E), generator=g, device=env.device, dtype=torch.float64) * 2 - 1).float().double()
prev = (0, 0, 0, 0)
import time
for t in range(STEPS):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, _, term, trunc, info = env.step(acts[t])
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    c = env.step_cycles().astype(np.float64)
    sub = info[:, 10].cpu().numpy()
    st = env.stragglers() + env.cycle_skips()
    top = np.argsort(-c)[:3]
    print("step %2d: %6.1f ms; slowest envs %s cycles %s sim steps %s; +STEP_LIMIT %d, +recurrences %d skipping %d sim steps" % (
        t, ms, top.tolist(), ["%.2e" % c[i] for i in top], [int(sub[i]) for i in top], st[1] - prev[1], st[2] - prev[2], st[3] - prev[3]))
    prev = st
    env.reset(term)
env.close()
