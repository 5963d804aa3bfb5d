"""The bench line on the device (-m gpu): a small run of bench.py must print ONE JSON line with the contract's keys, the roofline block, the two ceilings
measured in the run (roofline.ceiling from bp_get_cost_stats and the run's own clock, clock span beside the clock) and both throughput
accountings at the top level."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_small_ship_ice_line_has_the_contract_keys_and_the_ceilings():
    d = _bench("--envs-per-gpu", "256", "--trials", "8", "--steps", "4", "--warmup", "2", "--steady-steps", "3", "--no-cpu-baseline")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["dtype"] == "f64" and d["vs_baseline"] is None and d["value"] > 0
    assert d["steady_state_value"] == d["steady_state"]["value"] > 0 and "value_note" in d
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "physics_ms", "kernel", "clock_mhz", "clock_span_ms", "clock_per_xcd", "ceiling"):
        assert k in r, k
    assert r["kernel"] == "k_physics_step_schedl" and r["physics_ms"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = r["ceiling"]
    assert c["launches"] == 4 and c["wave_slots"] == 256          # 256 envs: one resident workgroup per env
    # both bounds are lower bounds of the launch, measured with the run's own clock: the kernel cannot be faster than its heaviest env, nor than all work over all slots
    assert 0 < c["work_over_slots_ms"] <= c["heaviest_chain_ms"] <= r["physics_ms"] * 1.02, c
    assert c["launch_over_chain"] >= 0.98 and c["launch_over_work"] >= c["launch_over_chain"]
    assert 1000 < r["clock_mhz"] < 3500 and r["clock_span_ms"] > 0
    assert any(x["xcd"] is not None and x["span_ms"] > 0 for x in r["clock_per_xcd"])
