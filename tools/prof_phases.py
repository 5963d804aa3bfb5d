"""Phase breakdown of k_physics from the BP_PROF diagnostic build: BP_PROF=1 python tools/prof_phases.py [E] [steps] [ship-ice|maze]"""
import os
import sys

os.environ["BP_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials

E = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 14
KIND = sys.argv[3] if len(sys.argv) > 3 else "ship-ice"   # ship-ice | maze
if KIND == "maze":
    from benchpush_amd.envs.maze_namo import BatchedMazeEnv
    env = BatchedMazeEnv(E, cfg={"num_obstacles": 20}, num_layouts=100, base_seed=0)
else:
    trials = default_trials(0.3, 100, base_seed=0)
    env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials)
env.reset()
prof = torch.zeros((E, 64), dtype=torch.int64, device=env.device)
env.L.bp_debug_prof(env.h, prof.data_ptr())
g = torch.Generator(device=env.device); g.manual_seed(1234)
names = ["integrate", "refresh", "cand+hint", "face_seps", "deliver", "filter", "prestep+warmset", "velint+warm", "solver", "post+mvlist", "manifolds"]
for t in range(STEPS):
    a = (torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1).float().double()
    prof.zero_()
    _, _, term, _, _ = env.step(a)
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.float64)
    if t >= STEPS - 3:
        tot = p[:, 23]
        worst = int(np.argmax(tot))
        print("step %d: kernel cycles(100MHz ticks?) mean %.0f max %.0f (env %d)" % (t, tot.mean(), tot.max(), worst))
        for who, row in (("mean", p.mean(0)), ("worst", p[worst])):
            print("  %s: " % who + " ".join("%s=%.1f%%" % (n, 100 * row[i] / row[23]) for i, n in enumerate(names)))
            print("        cycles per sub-step: " + " ".join("%s=%.0f" % (n, row[i] / 400) for i, n in enumerate(names)) + " total=%.0f" % (row[23] / 400))
            print("        per-substep: nmv=%.2f refresh=%.3f fullpairs=%.2f nact=%.2f levels=%.2f nwarm=%.2f" % (
                row[16] / 400, row[17] / 400, row[18] / 400, row[19] / 400, row[20] / 400, row[21] / 400))
            print("        narrow-phase stages, cycles per sub-step: " + " ".join("%s=%.0f" % (n, row[k] / 400) for n, k in (
                ("pose", 33), ("transform", 34), ("aabb", 0), ("drain", 35), ("candidates", 27), ("cached_planes", 28), ("bound_rounds", 29), ("plane_search+resolve", 30), ("normal", 31), ("support", 32))))
            print("        per-substep: cand_rounds=%.2f rounds_with_pairs=%.2f aabb_pairs=%.2f cached_plane_queries=%.2f search_batches=%.2f searched_planes=%.2f "
                  "support_queries=%.2f" % tuple(row[k] / 400 for k in (24, 25, 36, 37, 26, 38, 39)))
            print("        per-substep: warm_closure_iterations=%.2f recolourings=%.3f solver_iterations=%.2f colour_passes=%.2f passes_with_two_contacts=%.2f" %
                  tuple(row[k] / 400 for k in (40, 41, 42, 43, 44)))
    env.reset(term)
