"""Per-env wave cycles of k_bd_physics vs the launch (area-clearing / box-delivery): python tools/bd_cost_dist.py area|box [E] [steps]
and what a perfect dispatch order would give (list-scheduling model on the measured cycles)."""
import os, sys, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
kind = sys.argv[1] if len(sys.argv) > 1 else "area"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 16
if kind == "area":
    from benchpush_amd.envs.area_clearing import BatchedAreaClearingEnv
    env = BatchedAreaClearingEnv(E, num_trials=64)
else:
    from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
    env = BatchedBoxDeliveryEnv(E, cfg={"boxes": {"num_boxes_small": 12}}, num_trials=64)
env.reset()
g = torch.Generator(device=env.device); g.manual_seed(1234)


def sched(c, order, m=2048):
    h = [0.0] * m; heapq.heapify(h); end = 0.0
    for e in order:
        s0 = heapq.heappop(h); f = s0 + c[e]; heapq.heappush(h, f); end = max(end, f)
    return end


prev = None
for t in range(STEPS):
    a = torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1
    env.enable_timing(True)
    out = env.step(a)
    term, trunc = out[2], out[3]
    p_ms, _, _ = env.kernel_time_ms()
    c = env.step_cycles().astype(np.float64)
    if prev is not None and t >= 4:
        print("step %2d physics+finish %.1f ms | env Mcycles mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | sum/2048 %.1f | model: last step's order %.1f, oracle order %.1f, random %.1f | r(prev, now) %.2f" % (
            t, p_ms, c.mean() / 1e6, np.percentile(c, 50) / 1e6, np.percentile(c, 90) / 1e6, np.percentile(c, 99) / 1e6, c.max() / 1e6, c.sum() / 2048e6,
            sched(c, np.argsort(-prev, kind="stable")) / 1e6, sched(c, np.argsort(-c, kind="stable")) / 1e6, sched(c, np.random.default_rng(t).permutation(E)) / 1e6,
            np.corrcoef(prev, c)[0, 1]))
    prev = c
    env.reset((term | trunc) if kind == "area" else term)
