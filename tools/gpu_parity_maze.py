"""GPU-vs-oracle parity probe for maze-NAMO-v0 (run on the GPU box): python tools/gpu_parity_maze.py [E] [steps] [nbox]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from benchpush_amd.envs.maze_namo import BatchedMazeEnv
from oracle.oracle import OracleMaze

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
NBOX = int(sys.argv[3]) if len(sys.argv) > 3 else 20
T = 3
env = BatchedMazeEnv(E, cfg={"num_obstacles": NBOX}, num_layouts=T, base_seed=0)
obs, info = env.reset()
torch.cuda.synchronize()
env.check_errors()
cfg = env.cfg
orcs = [OracleMaze(env.params, cfg.robot.vertices, cfg.robot.wheel_vertices, cfg.obstacle_size) for _ in range(E)]
eps = [0] * E
ok = True
for e, o in enumerate(orcs):
    oo = o.reset(env.layouts[e % T])
    g = obs[e].cpu().numpy()
    if not np.array_equal(g, oo):
        d = np.argwhere(g != oo)
        print("reset obs mismatch env", e, len(d), d[:4].tolist(), g[tuple(d[0])], oo[tuple(d[0])])
        ok = False
print("nb_cap", env.nb_cap, "bodies", env.num_bodies()[:4], "oracle shapes", orcs[0].ns)


def cmp_state(tag):
    good = True
    bs = env.body_state().cpu().numpy()
    for e, o in enumerate(orcs):
        ss = o.shape_states()
        g = bs[e, : len(ss)]
        if not np.array_equal(g, ss):
            bad = np.argwhere(g != ss)
            print(tag, "env", e, "state mismatch", len(bad), bad[:5].tolist(), "maxabs", np.abs(g - ss).max())
            good = False
    return good


ok = cmp_state("reset") and ok
print("reset parity:", ok)
rng = np.random.default_rng(0)
for t in range(STEPS):
    a = rng.uniform(-1, 1, size=E)
    obs, rew, term, trunc, info = env.step(torch.from_numpy(a))
    torch.cuda.synchronize()
    env.check_errors()
    outs = [o.step(float(a[e])) for e, o in enumerate(orcs)]
    okb = cmp_state("step %d" % t)
    go, gi, gr, gt = obs.cpu().numpy(), info.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
    oko = True
    for e in range(E):
        oo, orr, ot, oi = outs[e]
        if not np.array_equal(go[e], oo):
            d = np.argwhere(go[e] != oo)
            print("step", t, "obs mismatch env", e, len(d), d[:4].tolist(), go[e][tuple(d[0])], oo[tuple(d[0])])
            oko = False
        oiv = np.array(list(oi.values()))
        if not np.array_equal(gi[e], oiv):
            print("step", t, "info mismatch env", e, gi[e], oiv)
            oko = False
        if gr[e] != orr or bool(gt[e]) != ot:
            print("step", t, "reward/term mismatch", e, gr[e], orr, gt[e], ot)
            oko = False
    print("step", t, "bodies", okb, "outputs", oko, "contacts", [int(o[3]["n_contact_pts"]) for o in outs][:6], "term", gt.tolist()[:6])
    for e in range(E):
        if outs[e][2]:
            m = torch.zeros(E, dtype=torch.uint8)
            m[e] = 1
            env.reset(m)
            eps[e] += 1
            orcs[e].reset(env.layouts[(e + eps[e]) % T])
