"""Summarise rocprofv3 csv outputs (kernel stats + PMC passes) into a short text report + pmc_traffic.json."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
build = sys.argv[2] if len(sys.argv) > 2 else "unknown"


def rows(pattern):
    """csv rows; of a kernel that is launched with several grid sizes (k_physics_step_sched: the scheduled launch and the completion launch that
    follows it, a few hundred workgroups that leave at once) only the dispatches with the largest grid are kept."""
    allr = []
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            allr.extend(csv.DictReader(fh))
    big = {}
    for r in allr:
        g = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
        k = r.get("Kernel_Name", "")
        big[k] = max(big.get(k, 0), g)
    for r in allr:
        g = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
        if "Kernel_Name" in r and g and g < big[r["Kernel_Name"]] and r["Kernel_Name"].startswith("k_physics_step_sched"):
            continue
        # with resident wavefronts (k_physics_step_schedl) every dispatch of k_physics_step_sched itself is a completion launch (at most 256 workgroups)
        if "Kernel_Name" in r and g and g <= 256 * 64 and r["Kernel_Name"].split("(")[0] == "k_physics_step_sched":
            continue
        yield r


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for r in rows("stats/**/*kernel_stats.csv"):
    print("%-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (r.get("Name", "")[:60], r.get("Calls"), r.get("TotalDurationNs"),
                                                          r.get("AverageNs"), r.get("Percentage")))
# per-kernel durations from the trace itself
dur = defaultdict(list)
for r in rows("stats/**/*kernel_trace.csv"):
    dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
# bench.py's timed launches: [W, W+K) for `value`, the last KS launches of the trace for `steady_state`
_tr = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows("stats/**/*kernel_trace.csv") if r["Kernel_Name"].startswith("k_physics_step"))
try:
    _b = json.load(open(os.path.join(out, "bench_under_rocprof.json")))
    W_, K_ = _b["warmup"], _b["steps"]
    seg = _tr[W_:W_ + K_]
    print("k_physics_step, launches %d..%d of the trace (the timed region of bench.py): avg_ms=%.3f; bench.py's own HIP-event time for them "
          "(roofline.physics_ms): %.3f" % (W_, W_ + K_ - 1, sum(b - a for a, b in seg) / len(seg) / 1e6, _b["roofline"]["physics_ms"]))
    if "steady_state" in _b:
        KS_ = _b["steady_state"]["steps"]
        seg = _tr[-KS_:]
        print("k_physics_step, last %d launches of the trace (steady_state region): avg_ms=%.3f; bench.py's HIP-event time for them "
              "(steady_state.physics_ms): %.3f" % (KS_, sum(b - a for a, b in seg) / len(seg) / 1e6, _b["steady_state"]["physics_ms"]))
    _ob = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows("stats/**/*kernel_trace.csv") if r["Kernel_Name"].startswith("k_observe"))
    # k_observe runs once per step over all envs and once per reset over the masked ones: the per-step launch is the first one that
    # starts after each k_physics_step
    big = []
    for a, b in _tr[W_:W_ + K_]:
        nxt = [y - x for x, y in _ob if x >= b]
        if nxt:
            big.append(nxt[0])
    if big:
        print("k_observe, the launch after each timed k_physics_step: n=%d avg_ms=%.4f (bench.py raster_kernel.ms: %.4f)" % (len(big), sum(big) / len(big) / 1e6, _b["roofline"]["raster_kernel"]["ms"]))
except Exception as e:
    print("bench json not usable:", e)
print("== kernel trace ==")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-50s n=%d avg_ms=%.3f min_ms=%.3f max_ms=%.3f" % (k[:50], len(v), sum(v) / len(v) / 1e6, min(v) / 1e6, max(v) / 1e6))
res = {}
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    acc = defaultdict(list)
    for r in rows(sub + "/**/*counter_collection.csv"):
        if r.get("Counter_Name") == name:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%s %-40s n=%d mean=%.1f KB per launch" % (name, k[:40], len(v), sum(v) / len(v)))
        res.setdefault(k, {})[name] = sum(v) / len(v)
sq = defaultdict(lambda: defaultdict(list))
for sub in ("pmc_sq", "pmc_sq2", "pmc_lds"):
    for r in rows(sub + "/**/*counter_collection.csv"):
        sq[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sq.items():
    print("SQ %-40s " % k[:40] + " ".join("%s=%.4g" % (c, sum(v[-6:]) / len(v[-6:])) for c, v in sorted(d.items())))
pmc = {"build": build, "envs": 4096, "kernel": "k_physics_step", "substeps_per_step": 400,
       "how": "rocprofv3 --kernel-trace --pmc, separate passes, python3 bench.py --steps 6 --warmup 24 (means of the last 6 launches: "
              "episodes 24-30 steps old, close to the steady-state mix)"}
for k, d in res.items():
    if k.startswith("k_physics_step") and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        # MI355X_MICROARCH.md HBM section: FETCH_SIZE / WRITE_SIZE in KB; on gfx950 FETCH_SIZE reads 1/2 of the bytes of a wide
        # coalesced stream -> doubled as prescribed (this kernel's narrow accesses are otherwise uncalibrated)
        pmc["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        pmc["fetch_kb_raw"], pmc["write_kb_raw"] = d["FETCH_SIZE"], d["WRITE_SIZE"]
        print("k_physics HBM bytes/launch (corrected): %.3e" % pmc["hbm_bytes_per_launch"])
for k, d in sq.items():
    if k.startswith("k_physics_step"):
        per = {c: sum(v[-6:]) / len(v[-6:]) for c, v in d.items()}
        pmc["per_launch"] = per
        pmc["kernel"] = k
        tot = sum(per.get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
        pmc["wave_instructions_per_env_substep"] = tot / (4096 * 400.0)
        pmc["per_env_substep"] = {c: per.get(c, 0.0) / (4096 * 400.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")}
        if per.get("SQ_ACTIVE_INST_VALU") and per.get("SQ_THREAD_CYCLES_VALU"):
            pmc["lanes_active"] = per["SQ_THREAD_CYCLES_VALU"] / (64.0 * per["SQ_ACTIVE_INST_VALU"])
        if per.get("SQ_WAVE_CYCLES"):
            pmc["wave_time_split"] = {"issuing": per.get("SQ_ACTIVE_INST_ANY", 0) / per["SQ_WAVE_CYCLES"],
                                      "waitcnt": per.get("SQ_WAIT_ANY", 0) / per["SQ_WAVE_CYCLES"],
                                      "issue_stall": per.get("SQ_WAIT_INST_ANY", 0) / per["SQ_WAVE_CYCLES"]}
per = pmc.get("per_launch", {})
if per.get("SQ_LDS_IDX_ACTIVE"):
    # SURVEY 8d's LDS line (MI355X_MICROARCH.md, LDS): SQ_LDS_BANK_CONFLICT = extra LDS-array cycles spent on conflicts, SQ_LDS_IDX_ACTIVE = all LDS-array
    # cycles; SQ_WAIT_INST_LDS = wave quad-cycles stalled at LDS *issue* (a sub-bucket of WAIT_INST_ANY, not the lgkmcnt drain); the LDS array's
    # own occupancy = its active cycles / (256 CUs x kernel cycles), from the kernel time of the stats pass
    pmc["lds"] = {"bank_conflict_frac": per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_LDS_IDX_ACTIVE"],
                  "addr_conflict_cycles": per.get("SQ_LDS_ADDR_CONFLICT"),
                  "array_cycles_per_env_substep": per["SQ_LDS_IDX_ACTIVE"] / (4096 * 400.0),
                  "cycles_per_lds_instruction": per["SQ_LDS_IDX_ACTIVE"] / max(per.get("SQ_INSTS_LDS", 1.0), 1.0),
                  "issue_stall_share_of_wave_time": (per.get("SQ_WAIT_INST_LDS", 0.0) / per["SQ_WAVE_CYCLES"]) if per.get("SQ_WAVE_CYCLES") else None,
                  "lds_issue_share_of_wave_time": (per.get("SQ_ACTIVE_INST_LDS", 0.0) / per["SQ_WAVE_CYCLES"]) if per.get("SQ_WAVE_CYCLES") else None}
    try:
        kms = [d for k, d in dur.items() if k.startswith("k_physics_step")][0]
        kcyc = (sum(kms) / len(kms)) * 1e-9 * 2.4e9
        pmc["lds"]["array_busy_frac"] = per["SQ_LDS_IDX_ACTIVE"] / (256.0 * kcyc)
    except Exception:
        pass
json.dump(pmc, open(os.path.join(out, "pmc.json"), "w"), indent=1)
print(json.dumps(pmc))
