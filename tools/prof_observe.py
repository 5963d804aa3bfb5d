"""Phase stamps of k_observe (BP_PROF build): BP_PROF=1 python tools/prof_observe.py [E] [steps]"""
import os, sys
os.environ["BP_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
trials = default_trials(0.3, 100, base_seed=0)
env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials)
env.reset()
g = torch.Generator(device=env.device); g.manual_seed(1234)
for t in range(STEPS):
    a = (torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1).float().double()
    env.step(a)
prof = torch.zeros((E, 64), dtype=torch.int64, device=env.device)
env.L.bp_debug_prof(env.h, prof.data_ptr())
env.observe()
torch.cuda.synchronize()
p = prof.cpu().numpy().astype(np.float64)
names = ["clear", "pretest+ship+edt", "load verts", "exact test", "rows", "footprint+line", "compose"]
p = p[:, 56:64]   # k_observe stamps live in slots 56..63 of the row
d = np.diff(p[:, :8], axis=1)
print("k_observe phases, cycles per workgroup (mean / p90): " + "; ".join("%s %.0f / %.0f" % (n, d[:, i].mean(), np.percentile(d[:, i], 90)) for i, n in enumerate(names)))
print("total %.0f / %.0f; span of the launch %.0f" % ((p[:, 7] - p[:, 0]).mean(), np.percentile(p[:, 7] - p[:, 0], 90), p[:, 7].max() - p[:, 0].min()))
