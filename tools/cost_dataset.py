"""Dataset for a per-env cost predictor (BP_PRED build): features at the end of step t (+ the action of step t+1) -> wave cycles of step t+1.
BP_PROF=1 BP_PROF_LIB=benchpush_amd/libbenchpush_hip_pred.so python tools/cost_dataset.py [E] [warm] [steps] [out.npz]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
WARM = int(sys.argv[2]) if len(sys.argv) > 2 else 30
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 30
OUT = sys.argv[4] if len(sys.argv) > 4 else "gpurun_out/cost_dataset.npz"
trials = default_trials(0.3, 100, base_seed=0)
env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials)
env.reset()
prof = torch.zeros((E, 64), dtype=torch.int64, device=env.device)
env.L.bp_debug_prof(env.h, prof.data_ptr())
g = torch.Generator(device=env.device); g.manual_seed(1234)
nb = torch.as_tensor(env.num_bodies(), device=env.device)


def features():
    """geometry of the field around the ship at the end of a step"""
    bs = env.body_state()                                  # [E, nbcap, 9]
    n = torch.as_tensor(env.num_bodies(), device=env.device)
    idx = torch.arange(bs.shape[1], device=env.device)[None, :]
    valid = (idx >= 1) & (idx < n[:, None])
    sx, sy, sa = bs[:, 0, 0], bs[:, 0, 1], bs[:, 0, 2]
    svx, svy, sw = bs[:, 0, 3], bs[:, 0, 4], bs[:, 0, 5]
    dx, dy = bs[:, :, 0] - sx[:, None], bs[:, :, 1] - sy[:, None]
    c, s = torch.cos(sa)[:, None], torch.sin(sa)[:, None]
    fwd, lat = dx * c + dy * s, -dx * s + dy * c           # ship frame: forward along the heading
    spd = torch.sqrt(bs[:, :, 3] ** 2 + bs[:, :, 4] ** 2)
    moving = valid & ((spd > 0) | (bs[:, :, 5] != 0))
    f = [sx, sy, sa, torch.sqrt(svx ** 2 + svy ** 2), sw, moving.sum(1).double(), (spd * valid).sum(1)]
    for (f0, f1, l1) in [(-1.5, 1.5, 1.5), (1.5, 3.0, 1.5), (3.0, 4.5, 1.5), (-1.5, 1.5, 3.0), (1.5, 3.0, 3.0), (3.0, 4.5, 3.0), (-3.0, 6.0, 4.5)]:
        inbox = valid & (fwd >= f0) & (fwd < f1) & (lat.abs() < l1)
        f.append(inbox.sum(1).double()); f.append((inbox & moving).sum(1).double())
    return torch.stack(f, 1)


X, Y, M = [], [], []
prev_feat = None
for t in range(WARM + STEPS):
    a = (torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1).float().double()
    _, _, term, _, _ = env.step(a)
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.float64)
    if prev_feat is not None and t > WARM:
        X.append(np.concatenate([prev_feat, a.cpu().numpy()[:, None]], 1)); Y.append(p[:, 8].copy())
    env.reset(term)
    geo = features().cpu().numpy()
    termn = term.cpu().numpy().astype(np.float64)
    prev_feat = np.concatenate([p[:, :9], geo, termn[:, None]], 1)
np.savez_compressed(OUT, X=np.stack(X).astype(np.float32), Y=np.stack(Y).astype(np.float32))
print("saved", OUT, np.stack(X).shape)
