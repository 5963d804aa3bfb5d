"""Phase breakdown of k_bd_physics (BP_PROF build): BP_PROF=1 python tools/prof_phases_bd.py [E] [steps]"""
import os
import sys

os.environ["BP_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv

E = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
env = BatchedBoxDeliveryEnv(E, cfg={"boxes": {"num_boxes_small": 12}}, num_trials=32)
env.reset()
prof = torch.zeros((E, 64), dtype=torch.int64, device=env.device)
env.L.bp_debug_prof(env.h, prof.data_ptr())
g = torch.Generator(device=env.device); g.manual_seed(1234)
names = ["integrate", "refresh", "cand+hint", "face_seps", "deliver", "filter", "prestep+warmset", "velint+warm", "solver", "post+mvlist", "manifolds"]
for t in range(STEPS):
    a = torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1
    _, _, term, _, _ = env.step(a)
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.float64)
    sub = p[:, 22]
    tot = p[:, 23]
    print("step %d: substeps mean %.0f max %.0f; cycles/substep mean %.0f; kernel cycles mean %.3g max %.3g" % (t, sub.mean(), sub.max(), (tot / sub).mean(), tot.mean(), tot.max()))
    row = p.sum(0)
    print("   " + " ".join("%s=%.1f%%" % (n, 100 * row[i] / row[23]) for i, n in enumerate(names)) + " control=%.1f%%" % (100 * row[15] / row[23]))
    print("   per-substep: nmv=%.2f refresh=%.4f fullpairs=%.2f nact=%.2f levels=%.2f nwarm=%.2f" % tuple(row[k] / row[22] for k in (16, 17, 18, 19, 20, 21)))
    w = int(np.argmax(tot))
    r = p[w]
    print("   worst env %d: substeps %.0f cycles/substep %.0f: " % (w, r[22], r[23] / r[22]) + " ".join("%s=%.1f%%" % (n, 100 * r[i] / r[23]) for i, n in enumerate(names)) + " control=%.1f%%" % (100 * r[15] / r[23]))
    print("      per-substep: nmv=%.2f refresh=%.4f fullpairs=%.2f nact=%.2f levels=%.2f nwarm=%.2f" % tuple(r[k] / r[22] for k in (16, 17, 18, 19, 20, 21)))
    print("      worst env, cycles per sub-step: " + " ".join("%s=%.0f" % (n, r[i] / r[22]) for i, n in enumerate(names)) + " control=%.0f" % (r[15] / r[22]))
    print("      worst env, stages: " + " ".join("%s=%.0f" % (n, r[k] / r[22]) for n, k in (("pose", 33), ("transform", 34), ("aabb", 0), ("drain", 35), ("candidates", 27),
          ("cached_planes", 28), ("bound_rounds", 29), ("plane_search", 30), ("normal", 31), ("support", 32))))
    print("      worst env per-substep: cand_rounds=%.2f aabb_pairs=%.2f cached_plane_queries=%.2f searched_planes=%.2f support_queries=%.2f solver_iterations=%.2f colour_passes=%.2f" %
          tuple(r[k] / r[22] for k in (24, 36, 37, 38, 39, 42, 43)))
    cps = tot / sub
    print("   cycles/substep percentiles 50/90/99/max: %.0f %.0f %.0f %.0f; total cycles percentiles: %.3g %.3g %.3g %.3g" % (
        np.percentile(cps, 50), np.percentile(cps, 90), np.percentile(cps, 99), cps.max(), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max()))
    env.reset(term)
