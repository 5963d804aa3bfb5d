"""GPU vs oracle parity for area-clearing-v0: python tools/gpu_parity_ac.py [E] [steps] [env] [action_type]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from benchpush_amd import area_clearing_scenario as A
from benchpush_amd.config import default_cfg
from benchpush_amd.envs.area_clearing import BatchedAreaClearingEnv
from oracle.oracle_bd import AC_INFO_KEYS, OracleAreaClearing

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = default_cfg("area_clearing")
if len(sys.argv) > 3:
    cfg.env = sys.argv[3]
if len(sys.argv) > 4:
    cfg.agent.action_type = sys.argv[4]
trials = A.generate_trials(cfg, 8)
t0 = time.time()
env = BatchedAreaClearingEnv(E, cfg={"env": cfg.env, "agent": {"action_type": cfg.agent.action_type}}, trials=trials)
print("load %.2fs nb_cap %d" % (time.time() - t0, env.nb_cap))
oracles = []
for e in range(E):
    o = OracleAreaClearing(A.area_clearing_physics_params(cfg), A.area_clearing_params(cfg), cfg)
    o.reset(trials[e % len(trials)], observe=False)
    oracles.append(o)
m, om = env.maps(0), oracles[0].maps()
d = m["dims"]; si, sj, SH, SW = int(d[4]), int(d[5]), int(d[2]), int(d[3])
win = (slice(si, si + SH), slice(sj, sj + SW))
print("maps: cspace", np.array_equal(m["cspace"], om["cspace"][win]), "recept", np.array_equal(m["recept"], om["recept"][win]), "small", np.array_equal(m["small_free"], om["small_free"]))
obs, info = env.reset()
torch.cuda.synchronize()


def compare(tag):
    st = env.body_state().cpu().numpy()
    ok = True
    for e in range(E):
        ost = oracles[e].shape_states()
        n = 6 + env.nbox
        if not np.array_equal(st[e, :n], ost[:n]):
            bad = np.argwhere((st[e, :n] != ost[:n]).any(1)).ravel()
            print(tag, "env", e, "BODY MISMATCH slots", bad[:8], "max abs", np.abs(st[e, :n] - ost[:n]).max()); ok = False
    return ok


print("reset parity:", compare("reset"))
oo = np.stack([o.observe() for o in oracles]); go = obs.cpu().numpy()
print("reset obs mismatches per channel:", [int((go[..., c] != oo[..., c]).sum()) for c in range(4)])
rng = np.random.RandomState(123)
for t in range(STEPS):
    if cfg.agent.action_type == "velocity":
        a = rng.uniform(-1, 1, (E, 2))
    elif cfg.agent.action_type == "position":
        a = rng.randint(0, 224 * 224, E).astype(np.float64)
    else:
        a = rng.uniform(-1, 1, E)
    t1 = time.time()
    obs, rew, term, trunc, info = env.step(torch.tensor(a))
    torch.cuda.synchronize()
    dt = time.time() - t1
    res = [o.step(a[e]) for e, o in enumerate(oracles)]
    oi = np.array([[r[4][k] for k in AC_INFO_KEYS] for r in res]); gi = info.cpu().numpy()
    oo = np.stack([r[0] for r in res]); go = obs.cpu().numpy()
    print("step", t, "gpu_s %.4f" % dt, "bodies", compare("step %d" % t), "info", np.array_equal(gi, oi), "reward", np.array_equal(rew.cpu().numpy(), np.array([r[1] for r in res])),
          "term", np.array_equal(term.cpu().numpy().astype(bool), np.array([r[2] for r in res])), "obs mism", [int((go[..., c] != oo[..., c]).sum()) for c in range(4)],
          "substeps", gi[:, 10].astype(int).tolist()[:6], "count", gi[:, 7].astype(int).tolist()[:6])
    if not np.array_equal(gi, oi):
        for (e, k) in np.argwhere(gi != oi)[:6]:
            print("   info env", e, AC_INFO_KEYS[k], gi[e, k], oi[e, k])
try:
    env.check_errors(); print("no capacity errors")
except Exception as ex:
    print("ERRORS:", ex)
