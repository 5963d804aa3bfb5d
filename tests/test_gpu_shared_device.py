"""A device that is not ours alone (-m gpu): resident step kernels hold every wave slot for a whole launch, which suits one process per GPU only.
Handles find out by themselves -- a shared advisory lock per PCI bus id, re-checked every 64 launches (csrc/bp_capi.hip: resident_*) -- and launch the dispatcher-driven
kernels while another PROCESS has a resident handle on the same device; results are identical either way."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import torch
from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
trials = default_trials(0.1, 2, base_seed=3)
env = BatchedShipIceEnv(8, cfg={"concentration": 0.1}, trials=trials, device="cuda:0")
print("child shared=%%d resident=%%d" %% (env.L.bp_device_shared(env.h), env.L.bp_sched_resident(env.h)), flush=True)
env.reset()
a = torch.zeros(8, dtype=torch.float64)
env.step(a)
torch.cuda.synchronize()
print("child stepped", flush=True)
sys.stdin.readline()            # stay alive (holding the lock) until the parent says so
env.close()
print("child closed", flush=True)
'''


def test_a_second_process_on_the_device_turns_resident_launches_off_and_back_on(tmp_path, monkeypatch):
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
    monkeypatch.delenv("BP_SCHED_PERSIST", raising=False)
    monkeypatch.delenv("BP_PAIR_RESIDENT", raising=False)
    monkeypatch.setenv("BP_LOCK_DIR", str(tmp_path))           # a lock file of this test's own
    trials = default_trials(0.1, 2, base_seed=3)
    env = BatchedShipIceEnv(8, cfg={"concentration": 0.1}, trials=trials, device="cuda:0")
    ref = BatchedShipIceEnv(8, cfg={"concentration": 0.1}, trials=trials, device="cuda:0")      # a second handle of THIS process does not count as sharing
    assert env.L.bp_device_shared(env.h) == 0 and env.L.bp_sched_resident(env.h) == 8
    assert ref.L.bp_device_shared(ref.h) == 0 and ref.L.bp_sched_resident(ref.h) == 8
    assert any(f.startswith("benchpush_amd.resident.") for f in os.listdir(str(tmp_path)))
    env.reset(); ref.reset()
    g = np.random.default_rng(0)

    def step_both(n):
        for _ in range(n):
            a = torch.from_numpy(g.uniform(-1, 1, 8).astype(np.float32).astype(np.float64))
            o1, r1, t1, _, i1 = env.step(a)
            o2, r2, t2, _, i2 = ref.step(a)
            assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(i1, i2) and torch.equal(env.body_state(), ref.body_state())
            env.reset(t1); ref.reset(t2)

    step_both(3)
    script = tmp_path / "child.py"
    script.write_text(_CHILD % ROOT)
    p = subprocess.Popen([sys.executable, str(script)], env=dict(os.environ), stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        line = p.stdout.readline().decode()
        while line and not line.startswith("child shared"):
            line = p.stdout.readline().decode()
        assert line.strip() == "child shared=1 resident=0", line      # the newcomer sees the device taken and never launches a resident kernel
        while line and not line.startswith("child stepped"):
            line = p.stdout.readline().decode()
        assert line.startswith("child stepped")
        step_both(70)                                                  # the check runs every 64 launches of a handle
        assert env.L.bp_device_shared(env.h) == 1 and env.L.bp_sched_resident(env.h) == 0
        p.stdin.write(b"\n"); p.stdin.flush()
        out = p.stdout.read().decode()
        assert "child closed" in out, out
        assert p.wait(timeout=120) == 0
    finally:
        if p.poll() is None:
            p.kill()
    step_both(70)                                                      # alone again: back to resident wavefronts
    assert env.L.bp_device_shared(env.h) == 0 and env.L.bp_sched_resident(env.h) == 8
    env.check_errors(); ref.check_errors()
    env.close(); ref.close()


def test_explicit_switch_turns_the_check_off(monkeypatch, tmp_path):
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
    monkeypatch.setenv("BP_LOCK_DIR", str(tmp_path))
    monkeypatch.setenv("BP_SCHED_PERSIST", "1")
    env = BatchedShipIceEnv(4, cfg={"concentration": 0.1}, trials=default_trials(0.1, 2, base_seed=3), device="cuda:0")
    assert env.L.bp_sched_resident(env.h) == 4 and env.L.bp_device_shared(env.h) == 0
    assert not os.listdir(str(tmp_path))                               # no lock file: the user has decided
    env.close()
