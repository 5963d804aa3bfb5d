"""What would a perfect cost predictor buy?  Run the same deterministic trajectory twice; the second time every step is dispatched in the order of
its own measured per-env cycles (bp_set_step_cost_hint).  python tools/oracle_order.py [E] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
trials = default_trials(0.3, 100, base_seed=0)


def run(hints=None, mode=None):
    env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials)
    env.reset()
    g = torch.Generator(device=env.device); g.manual_seed(1234)
    rng = np.random.default_rng(7)
    costs, ms = [], []
    for t in range(STEPS):
        a = (torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1).float().double()
        if t > 0:
            if mode == "oracle": env.set_cost_hint((hints[t] / 256).astype(np.uint32))
            elif mode == "random": env.set_cost_hint(rng.integers(0, 1 << 20, E).astype(np.uint32))
            elif mode == "blend": env.set_cost_hint(((0.5 * hints[t] + 0.5 * hints[t - 1]) / 256).astype(np.uint32))
        env.enable_timing(True)
        _, _, term, _, _ = env.step(a)
        p_ms, _, _ = env.kernel_time_ms()
        costs.append(env.step_cycles().astype(np.float64)); ms.append(p_ms)
        env.reset(term)
    env.close()
    return np.array(costs), np.array(ms)


cA, msA = run()
cB, msB = run(cA, "oracle")
cC, msC = run(cA, "random")
cD, msD = run(cA, "blend")
s = slice(STEPS // 2, STEPS)
print("physics ms per step, steps %d..%d: default order (last step's cycles) %.2f | oracle order %.2f | random order %.2f | 50/50 blend of oracle and last step %.2f" % (
    s.start, s.stop, msA[s].mean(), msB[s].mean(), msC[s].mean(), msD[s].mean()))
print("max chain M cycles: default %.1f oracle-run %.1f; sum/2048: %.1f / %.1f" % (cA[s].max(1).mean() / 1e6, cB[s].max(1).mean() / 1e6, cA[s].sum(1).mean() / 2048e6, cB[s].sum(1).mean() / 2048e6))
