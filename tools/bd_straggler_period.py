"""Do the 10 000-sim-step stragglers of box-delivery revisit a state exactly?   python tools/bd_straggler_period.py [E] [steps]

Pass 1 finds (env step, env) pairs whose k_bd_physics chain is far above the rest; pass 2 replays the same seeded run with bp_debug_trace on that env
and looks for the smallest p with pose[s] == pose[s - p] (bitwise, every body) over the last 2000 sim steps of the step."""
import ctypes as C
import os
import sys

# bp_debug_trace lives in the diagnostic twin of the library (python -c "from benchpush_amd.build import build_debug_paths; build_debug_paths()")
os.environ.setdefault("BP_PROF", "1")
os.environ.setdefault("BP_PROF_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchpush_amd", "libbenchpush_hip_dbgpaths.so"))

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
CAP = 10100


def run(trace=None):
    env = BatchedBoxDeliveryEnv(E, cfg={"boxes": {"num_boxes_small": 12}}, num_trials=64)
    env.reset()
    nbcap = env.L.bp_nb_cap(env.h) if hasattr(env.L, "bp_nb_cap") else 64
    g = torch.Generator(device=env.device); g.manual_seed(1234)
    found = []
    for t in range(STEPS):
        a = torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1
        buf = None
        if trace is not None and trace[0] == t:
            buf = torch.full((CAP, nbcap, 3), float("nan"), dtype=torch.float64, device=env.device)
            rc = env.L.bp_debug_trace(env.h, C.c_void_p(buf.data_ptr()), int(trace[1]))
            if rc != 0:
                raise SystemExit("bp_debug_trace: rc %d -- this tool needs the -DBP_DEBUG_PATHS twin of the library (benchpush_amd.build.build_debug_paths)" % rc)
        _, _, term, _, _ = env.step(a)
        torch.cuda.synchronize()
        if buf is not None:
            env.L.bp_debug_trace(env.h, None, 0)
            return buf.cpu().numpy()
        c = env.step_cycles().astype(np.float64)
        if c.max() > 2.2e8:
            found.append((t, int(np.argmax(c)), c.max()))
        env.reset(term)
    return found


found = run()
print("stragglers (env step, env, cycles):", found)
for t, e, cyc in found[:4]:
    tr = run((t, e))
    n = int(np.isfinite(tr[:, 0, 0]).sum())
    bits = tr[:n].view(np.uint64).reshape(n, -1)
    live = ~np.isnan(tr[n - 1]).any(axis=1)
    print("step %d env %d: %d sim steps, %.3g cycles, %d bodies" % (t, e, n, cyc, int(live.sum())))
    moved = (bits[n - 1] != bits[n - 2001]).reshape(-1, 3).any(axis=1)
    print("   bodies whose pose differs between sim steps n-1 and n-2001:", np.nonzero(moved)[0].tolist())
    best = None
    for p in range(1, 1500):
        if np.array_equal(bits[n - 2000:n], bits[n - 2000 - p:n - p]):
            best = p
            break
    print("   exact period over the last 2000 sim steps:", best)
    if best:
        s = n - 1
        while s - best >= 0 and np.array_equal(bits[s], bits[s - best]):
            s -= 1
        print("   periodic from sim step", s + 1 - best)
    else:
        d = np.abs(tr[n - 1] - tr[n - 2])
        print("   max |pose change| in the last sim step: %.3g; distinct poses of the last 2000 steps: %d" % (np.nanmax(d), len({b.tobytes() for b in bits[n - 2000:n]})))
