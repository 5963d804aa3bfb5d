"""GPU-vs-oracle parity probe (run on the GPU box): python tools/gpu_parity_debug.py [E] [steps] [conc]"""
import os
import sys

# bp_debug_trace lives in the diagnostic twin of the library (python -c "from benchpush_amd.build import build_debug_paths; build_debug_paths()")
os.environ.setdefault("BP_PROF", "1")
os.environ.setdefault("BP_PROF_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchpush_amd", "libbenchpush_hip_dbgpaths.so"))
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
from oracle.oracle import OracleShipIce

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
CONC = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
T = 4
trials = default_trials(CONC, T, base_seed=0)
print("floes per trial:", [len(t["obstacles"]) for t in trials])
env = BatchedShipIceEnv(E, cfg={"concentration": CONC}, trials=trials)
t0 = time.time()
obs, info = env.reset()
torch.cuda.synchronize()
print("reset time", time.time() - t0)
env.check_errors()
cfg = env.cfg
orcs = [OracleShipIce(env.params, cfg.ship.vertices, cfg.ship.head, cfg.ship.tail) for _ in range(E)]
oobs = []
for e, o in enumerate(orcs):
    ob, _ = o.reset(trials[e % T])
    oobs.append(ob)


def compare(tag):
    ok = True
    bs = env.body_state().cpu().numpy()
    nb = env.num_bodies()
    for e, o in enumerate(orcs):
        ob = o.bodies()
        if nb[e] != len(ob):
            print(tag, "env", e, "nb mismatch", nb[e], len(ob))
            ok = False
            continue
        g = bs[e, : nb[e]]
        if not np.array_equal(g, ob):
            bad = np.argwhere(g != ob)
            print(tag, "env", e, "body mismatch count", len(bad), "first", bad[:5].tolist(),
                  "maxabs", np.abs(g - ob).max())
            ok = False
    return ok


ok = compare("reset")
go = obs.cpu().numpy()
for e in range(E):
    if not np.array_equal(go[e], oobs[e]):
        d = np.argwhere(go[e] != oobs[e])
        print("reset obs mismatch env", e, len(d), d[:5].tolist())
        ok = False
print("reset parity:", ok)
rng = np.random.default_rng(0)
for t in range(STEPS):
    a = rng.uniform(-1, 1, size=E).astype(np.float32).astype(np.float64)
    trace = None
    if os.environ.get("BP_TRACE"):
        trace = torch.zeros((env.params["steps"], env.nb_cap, 3), dtype=torch.float64, device=env.device)
        env.debug_trace(trace, int(os.environ.get("BP_TRACE_ENV", "0")))
    t0 = time.time()
    obs, rew, term, trunc, info = env.step(torch.from_numpy(a))
    torch.cuda.synchronize()
    dt = time.time() - t0
    env.check_errors()
    outs = [o.step(float(a[e])) for e, o in enumerate(orcs)]
    okb = compare("step %d" % t)
    go = obs.cpu().numpy()
    gi = info.cpu().numpy()
    gr = rew.cpu().numpy()
    gt = term.cpu().numpy()
    oko = True
    for e in range(E):
        oo, orr, ot, oi = outs[e]
        if not np.array_equal(go[e], oo):
            d = np.argwhere(go[e] != oo)
            print("step", t, "obs mismatch env", e, len(d), d[:6].tolist(), go[e][tuple(d[0])], oo[tuple(d[0])])
            oko = False
        oiv = np.array(list(oi.values()))
        if not np.array_equal(gi[e], oiv):
            print("step", t, "info mismatch env", e, gi[e], oiv)
            oko = False
        if gr[e] != orr or bool(gt[e]) != ot:
            print("step", t, "reward/term mismatch env", e, gr[e], orr, gt[e], ot)
            oko = False
    print("step", t, "gpu_s %.4f" % dt, "bodies", okb, "outputs", oko, "contacts", [int(o[3]["n_contact_pts"]) for o in outs][:8])
    for e in range(E):
        if outs[e][2]:
            m = torch.zeros(E, dtype=torch.uint8)
            m[e] = 1
            env.reset(m)
            ep = getattr(orcs[e], "_ep", 0) + 1
            orcs[e]._ep = ep
            orcs[e].reset(trials[(e + ep) % T])
