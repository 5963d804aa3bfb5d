"""Which end-of-step quantity predicts the NEXT step's per-env cost best?  (BP_PRED build: tools/build_variant.sh pred "-DBP_PRED=1")
BP_PROF=1 BP_PROF_LIB=benchpush_amd/libbenchpush_hip_pred.so python tools/cost_predictors.py [E] [warm] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
WARM = int(sys.argv[2]) if len(sys.argv) > 2 else 40
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 12
trials = default_trials(0.3, 100, base_seed=0)
env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials, auto_reset=True) if "auto_reset" in BatchedShipIceEnv.__init__.__code__.co_varnames else BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials)
env.reset()
prof = torch.zeros((E, 64), dtype=torch.int64, device=env.device)
env.L.bp_debug_prof(env.h, prof.data_ptr())
g = torch.Generator(device=env.device); g.manual_seed(1234)
names = ["proxy 400", "proxy last 200", "last 100", "last 50", "last 10", "nmv end", "nact end", "nslots end", "cycles"]
prev = None
acc = {}
sim = {}
for t in range(WARM + STEPS):
    a = (torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1).float().double()
    _, _, term, _, _ = env.step(a)
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.float64)
    c = p[:, 8]
    termn = term.cpu().numpy().astype(bool)
    if prev is not None and t >= WARM:
        top = np.argsort(-c)[:64]
        preds = {n: prev[:, i] for i, n in enumerate(names)}
        preds["0.5*proxy400 + 2*last100"] = 0.5 * prev[:, 0] + 2 * prev[:, 2]
        preds["proxy400 + 8*last50"] = prev[:, 0] + 8 * prev[:, 3]
        preds["last100 (reset envs -> 0)"] = np.where(prev_term, 0.0, prev[:, 2])
        preds["last50 (reset envs -> 0)"] = np.where(prev_term, 0.0, prev[:, 3])
        for n, v in preds.items():
            order = np.argsort(-v, kind="stable")
            rank = np.empty(E, int); rank[order] = np.arange(E)
            r = rank[top]
            # start-time model: position k of the dispatch order starts in round 1 if k < 2048
            acc.setdefault(n, []).append((np.mean(r < 256), np.mean(r < 1024), np.mean(r < 2048), np.corrcoef(v, c)[0, 1]))
    if prev is not None and t >= WARM:
        # greedy list scheduling of this step's measured per-env cycles on 2048 wave slots in the order each predictor would dispatch
        import heapq
        def sched(order):
            h = [0.0] * 2048
            heapq.heapify(h)
            end = 0.0
            for e in order:
                s0 = heapq.heappop(h); f = s0 + c[e]; heapq.heappush(h, f); end = max(end, f)
            return end
        cands = dict(preds); cands["oracle (this step's cycles)"] = c; cands["random"] = np.random.default_rng(t).random(E)
        for n, v in cands.items():
            sim.setdefault(n, []).append(sched(np.argsort(-v, kind="stable")) / 1e6)
        sim.setdefault("[max chain]", []).append(c.max() / 1e6); sim.setdefault("[sum / 2048]", []).append(c.sum() / 2048 / 1e6)
    prev = p.copy(); prev_term = termn
    env.reset(term)
print("predictor: share of the next step's 64 heaviest envs inside the predicted top-256 / top-1024 / top-2048 (first round); correlation with next cycles")
for n, v in acc.items():
    m = np.mean(v, axis=0)
    print("%-28s %.2f %.2f %.2f   r=%.2f" % (n, m[0], m[1], m[2], m[3]))
print("list-scheduling model: end of the launch in M cycles (mean over steps) when dispatching in the predictor's order")
for n, v in sim.items():
    print("%-28s %.1f" % (n, np.mean(v)))
