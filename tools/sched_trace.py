#!/usr/bin/env python3
"""Timeline of the preemptive step scheduler from the -DBP_SCHED_TRACE diagnostic build: every task (a run of one env between two parks) with its XCD and its
start / end in the 100 MHz reference clock.  Where do the wave slots idle, which envs end the launch, how even are the XCDs?
    tools/build_variant.sh schedtrace "-DBP_SCHED_TRACE=1" && BP_PROF=1 BP_PROF_LIB=benchpush_amd/libbenchpush_hip_schedtrace.so python tools/sched_trace.py [E] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
trials = default_trials(0.3, 100, base_seed=0)
env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials)
env.reset()
prof = torch.zeros((E, 64), dtype=torch.int64, device=env.device)
env.L.bp_debug_prof(env.h, prof.data_ptr())
g = torch.Generator(device=env.device); g.manual_seed(1234)
SLOTS = 2048
for t in range(STEPS):
    a = (torch.rand(E, generator=g, device=env.device, dtype=torch.float64) * 2 - 1).float().double()
    if os.environ.get("BP_TRACE_DET"):   # actions that depend on (env id, step) only: the first envs of runs with different E see the same episode
        a = torch.remainder(torch.arange(E, device=env.device, dtype=torch.float64) * 0.37 + t * 0.11, 2.0) - 1.0
    prof.zero_()
    _, _, term, _, _ = env.step(a)
    torch.cuda.synchronize()
    if t >= STEPS - 3:
        p = prof.cpu().numpy().astype(np.uint64).reshape(-1)
        n = int(p[0])
        rec = p[8:8 + 4 * n].reshape(n, 4)
        envs = (rec[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        lev_in = ((rec[:, 0] >> np.uint64(32)) & np.uint64(0xFF)).astype(int)
        lev_out = ((rec[:, 0] >> np.uint64(40)) & np.uint64(0xFF)).astype(int)
        home = ((rec[:, 0] >> np.uint64(48)) & np.uint64(0xF)).astype(int)
        first = ((rec[:, 0] >> np.uint64(56)) & np.uint64(1)).astype(int)
        t0 = rec[:, 1].astype(np.int64); t1 = rec[:, 2].astype(np.int64)
        pre = (rec[:, 3] >> np.uint64(32)).astype(np.int64) / 100.0          # microseconds between the workgroup's start and the task's start (queue pop, waiting)
        hwid = ((rec[:, 3] >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
        idle = (rec[:, 3] & np.uint64(0xFFFF)).astype(int)
        if int(p[7]) > 0 and int(p[4]) > 0:      # sums over the tasks that ended with a park (100 MHz ticks): [1] resume = everything before the first sub-step, [2] the first sub-step after a resume, [3] park, [6] / [5] mean sub-step
            print("   parked tasks %d: park %.1f us each; resumed ones %d: resume %.1f us, first sub-step after it %.1f us; mean sub-step of those runs %.1f us" % (
                int(p[7]), p[3] / 100.0 / int(p[7]), int(p[4]), p[1] / 100.0 / int(p[4]), p[2] / 100.0 / int(p[4]), p[6] / 100.0 / max(int(p[5]), 1)))
        T0 = t0.min(); t0 = (t0 - T0) / 100.0; t1 = (t1 - T0) / 100.0          # microseconds since the first task started
        end = t1.max()
        busy = (t1 - t0).sum()
        print("step %d: %d tasks (%.2f per env), launch %.2f ms, slot-time %.1f slot-ms = %.1f %% of %d slots x launch; work bound %.2f ms" % (
            t, n, n / E, end / 1e3, busy / 1e3, 100 * busy / (SLOTS * end), SLOTS, busy / SLOTS / 1e3))
        q = first == 0
        if q.any(): print("   tasks taken from the queues: %d; workgroup start -> task start: mean %.1f us, p50 %.1f, p90 %.1f, p99 %.1f, sum %.1f slot-ms; with no empty poll: %d tasks, mean %.1f us" % (
            q.sum(), pre[q].mean(), np.percentile(pre[q], 50), np.percentile(pre[q], 90), np.percentile(pre[q], 99), pre[q].sum() / 1e3, (q & (idle == 0)).sum(),
            pre[q & (idle == 0)].mean() if (q & (idle == 0)).any() else 0))
        # per hardware slot (XCD, SE / CU / SIMD / wave id): the gap between the end of a task and the start of the next one in the same slot
        slot = home * 65536 + hwid
        o_ = np.lexsort((t0, slot))
        ss, a0, a1, pr = slot[o_], t0[o_], t1[o_], pre[o_]
        same = ss[1:] == ss[:-1]
        gaps = (a0[1:] - pr[1:] - a1[:-1])[same]          # next workgroup's own start minus this task's end
        if same.any(): print("   %d distinct slots seen; gap end-of-task -> start of the next workgroup in the same slot: n %d mean %.1f us p50 %.1f p90 %.1f p99 %.1f max %.1f, sum %.1f slot-ms" % (
            len(np.unique(slot)), same.sum(), gaps.mean(), np.percentile(gaps, 50), np.percentile(gaps, 90), np.percentile(gaps, 99), gaps.max(), gaps.sum() / 1e3))
        # utilisation over time in 0.5 ms bins
        edges = np.arange(0, end + 500, 500.0)
        util = []
        for a0, a1 in zip(edges[:-1], edges[1:]):
            util.append(np.clip(np.minimum(t1, a1) - np.maximum(t0, a0), 0, None).sum() / (a1 - a0))
        print("   running tasks per 0.5 ms bin: " + " ".join("%d" % round(u) for u in util))
        # per XCD: busy time and the end of its last task
        for x in range(8):
            m = home == x
            print("   XCD %d: %5d tasks, slot-time %7.1f ms, last end %.2f ms, envs finished there %d" % (x, m.sum(), (t1[m] - t0[m]).sum() / 1e3, t1[m].max() / 1e3 if m.any() else 0, int((lev_out[m] == 255).sum())))
        # the envs that finish last: their run time, the time they spent waiting between tasks, their number of tasks
        done = lev_out == 255
        last = np.argsort(-t1 * done)[:12]
        print("   last finishers: " + "; ".join("env %d end %.2f ms run %.2f wait %.2f tasks %d" % (
            envs[i], t1[i] / 1e3, (t1[envs == envs[i]] - t0[envs == envs[i]]).sum() / 1e3,
            (t1[i] - (t1[envs == envs[i]] - t0[envs == envs[i]]).sum() - t0[envs == envs[i]].min()) / 1e3, int((envs == envs[i]).sum())) for i in last))
        # total run time per env: distribution
        order = np.argsort(envs, kind="stable")
        run = np.zeros(E); np.add.at(run, envs, t1 - t0)
        endt = np.zeros(E); np.maximum.at(endt, envs, t1)
        print("   run time per env (ms): mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f;  end time: p50 %.2f p90 %.2f p99 %.2f max %.2f" % (
            run.mean() / 1e3, np.percentile(run, 50) / 1e3, np.percentile(run, 90) / 1e3, np.percentile(run, 99) / 1e3, run.max() / 1e3,
            np.percentile(endt, 50) / 1e3, np.percentile(endt, 90) / 1e3, np.percentile(endt, 99) / 1e3, endt.max() / 1e3))
        if os.environ.get("BP_TRACE_DUMP") and t == STEPS - 1:
            np.save(os.environ["BP_TRACE_DUMP"], run)
    env.reset(term)
