"""GPU vs oracle parity for box-delivery-v0: python tools/gpu_parity_bd.py [E] [steps] [obstacle_config]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from benchpush_amd import box_delivery_scenario as S
from benchpush_amd.config import default_cfg
from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
from oracle.oracle_bd import OracleBoxDelivery

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = default_cfg("box_delivery")
if len(sys.argv) > 3:
    cfg.env.obstacle_config = sys.argv[3]
trials = S.generate_trials(cfg, 8)
t0 = time.time()
env = BatchedBoxDeliveryEnv(E, cfg={"env": {"obstacle_config": cfg.env.obstacle_config}}, trials=trials)
print("load %.2fs nb_cap %d" % (time.time() - t0, env.nb_cap))
# static maps
om = None
oracles = []
for e in range(E):
    o = OracleBoxDelivery(S.box_delivery_physics_params(cfg), S.box_delivery_params(cfg), cfg)
    o.reset(trials[e % len(trials)], observe=False)
    oracles.append(o)
m = env.maps(0)
om = oracles[0].maps()
d = m["dims"]; si, sj, SH, SW = int(d[4]), int(d[5]), int(d[2]), int(d[3])
print("maps: cspace", np.array_equal(m["cspace"], om["cspace"][si:si + SH, sj:sj + SW]), "thin", np.array_equal(m["cspace_thin"], om["cspace_thin"][si:si + SH, sj:sj + SW]),
      "edt", np.array_equal(m["edt"][..., 0].astype(int) + si, om["edt_i"][si:si + SH, sj:sj + SW]) and np.array_equal(m["edt"][..., 1].astype(int) + sj, om["edt_j"][si:si + SH, sj:sj + SW]),
      "recept", np.array_equal(m["recept"], om["recept"][si:si + SH, sj:sj + SW]), "small", np.array_equal(m["small_free"], om["small_free"]),
      "free outside window", int(om["cspace"].sum() - om["cspace"][si:si + SH, sj:sj + SW].sum()))
obs, info = env.reset()
torch.cuda.synchronize()
st = env.body_state().cpu().numpy()


def compare(tag):
    ok = True
    st = env.body_state().cpu().numpy()
    alive, wp, nwp = env.box_state()
    for e in range(E):
        ost = oracles[e].shape_states()
        nphys = 6 + env.nbox
        a = st[e, :nphys]; b = ost[:nphys]
        al = oracles[e].alive().astype(bool)
        sel = np.ones(nphys, bool); sel[6:6 + env.nbox] = al
        same = np.array_equal(a[sel], b[sel])
        if not same:
            bad = np.argwhere((a != b).any(1) & sel).ravel()
            print(tag, "env", e, "BODY MISMATCH slots", bad[:8], "max abs", np.abs(a[sel] - b[sel]).max())
            ok = False
        if not np.array_equal(alive[e, :env.nbox].astype(bool), al):
            print(tag, "env", e, "alive mismatch", alive[e, :env.nbox], al.astype(int)); ok = False
    return ok


print("reset parity:", compare("reset"))
oo = np.stack([o.observe() for o in oracles])
go = obs.cpu().numpy()
print("reset obs mismatches per channel:", [(go[..., c] != oo[..., c]).sum() for c in range(4)])
rng = np.random.RandomState(123)
for t in range(STEPS):
    a = rng.uniform(-1, 1, E)
    t1 = time.time()
    obs, rew, term, trunc, info = env.step(torch.tensor(a))
    torch.cuda.synchronize()
    dt = time.time() - t1
    res = [o.step(float(a[e])) for e, o in enumerate(oracles)]
    oi = np.array([[r[4][k] for k in __import__("oracle.oracle_bd", fromlist=["x"]).BD_INFO_KEYS] for r in res])
    gi = info.cpu().numpy()
    oo = np.stack([r[0] for r in res]); go = obs.cpu().numpy()
    okb = compare("step %d" % t)
    print("step", t, "gpu_s %.4f" % dt, "bodies", okb, "info", np.array_equal(gi, oi), "reward", np.array_equal(rew.cpu().numpy(), np.array([r[1] for r in res])),
          "term", np.array_equal(term.cpu().numpy().astype(bool), np.array([r[2] for r in res])), "obs mism", [(go[..., c] != oo[..., c]).sum() for c in range(4)],
          "substeps", gi[:, 10].astype(int).tolist()[:6])
    if not np.array_equal(gi, oi):
        bad = np.argwhere(gi != oi)
        for (e, k) in bad[:6]:
            print("   info env", e, __import__("oracle.oracle_bd", fromlist=["x"]).BD_INFO_KEYS[k], gi[e, k], oi[e, k])
        _, wp, nwp = env.box_state()
        for e in sorted(set(bad[:, 0]))[:2]:
            print("   wp gpu", wp[e, :nwp[e]].round(4).tolist(), "oracle", oracles[e].last_waypoints().round(4).tolist())
try:
    env.check_errors()
    print("no capacity errors")
except Exception as ex:
    print("ERRORS:", ex)
