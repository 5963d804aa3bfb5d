#!/usr/bin/env python3
"""Paired kernel (BP_PAIR=<mode>) against the one-env-per-wavefront kernel (BP_SCHED=0) on the same trials and actions, step by step: first divergence,
capacity flags, which bodies differ.    python tools/gpu_pair_debug.py E conc T steps seed [mode]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials  # noqa: E402

E, conc, T, steps, seed = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
mode = sys.argv[6] if len(sys.argv) > 6 else "1"
trials = default_trials(conc, T, base_seed=seed)


def mk(envv):
    for k in ("BP_PAIR", "BP_SCHED"):
        os.environ.pop(k, None)
    os.environ.update(envv)
    e = BatchedShipIceEnv(E, cfg={"concentration": conc}, trials=trials, device="cuda:0")
    e.reset()
    return e


ref, got = mk({"BP_SCHED": "0"}), mk({"BP_PAIR": mode})
print("pair mode", got.L.bp_pair_mode(got.h), "nb_cap", got.nb_cap)
rng = np.random.default_rng(seed)
for t in range(steps):
    a = torch.from_numpy(rng.uniform(-1, 1, E).astype(np.float32).astype(np.float64))
    o1, r1, t1, _, i1 = ref.step(a)
    o2, r2, t2, _, i2 = got.step(a)
    b1, b2 = ref.body_state().cpu().numpy(), got.body_state().cpu().numpy()
    bad = [e for e in range(E) if not np.array_equal(b1[e], b2[e])]
    errs = []
    for nm, env in (("ref", ref), ("pair", got)):
        try:
            env.check_errors()
        except Exception as ex:  # noqa: BLE001
            errs.append((nm, str(ex)[:120]))
    print("step", t, "envs differing:", bad[:16], "info equal:", bool(torch.equal(i1, i2)), "obs equal:", bool(torch.equal(o1, o2)), errs)
    for e in bad[:4]:
        d = np.nonzero((b1[e] != b2[e]).any(axis=1))[0]
        print("   env", e, "bodies", d[:12].tolist(), "max |d|", float(np.abs(b1[e] - b2[e]).max()), "info ref", i1[e, 11:].tolist(), "pair", i2[e, 11:].tolist())
    if bad:
        break
    ref.reset(t1)
    got.reset(t2)
