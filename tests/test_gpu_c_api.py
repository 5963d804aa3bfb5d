"""The C ABI driven by a plain C++ program (examples/c_api_demo.cpp: hipMalloc'ed buffers, no Python / torch in the process) must produce
the same bits as the Python host path on the same scenario tables and actions."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_plain_cpp_program_matches_python_path(tmp_path):
    from benchpush_amd import _lib
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
    from benchpush_amd.scenario import pack_trials
    E, T, steps = 6, 3, 5
    trials = default_trials(0.3, T, base_seed=31)
    env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials)
    pk = pack_trials(trials, max_verts=20)
    Tn, F, V = pk["verts"].shape[:3]
    rng = np.random.default_rng(2)
    actions = rng.uniform(-1, 1, (steps, E)).astype(np.float32).astype(np.float64)
    cfg_bytes = bytes(_lib.make_config(env.params, env.cfg.ship.vertices, env.cfg.ship.head, env.cfg.ship.tail))
    blob = tmp_path / "scenario.bin"
    with open(blob, "wb") as f:
        f.write(np.array([E, Tn, F, V, steps], np.int32).tobytes())
        f.write(cfg_bytes)
        for k, dt in (("verts", np.float64), ("counts", np.int32), ("centres", np.float64), ("starts", np.float64), ("nfloes", np.int32)):
            f.write(np.ascontiguousarray(pk[k], dt).tobytes())
        f.write(actions.tobytes())
    exe = tmp_path / "c_api_demo"
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    libdir = os.path.join(ROOT, "benchpush_amd")
    subprocess.check_call([hipcc, "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_api_demo.cpp"), "-L" + libdir,
                           "-lbenchpush_hip", "-Wl,-rpath," + libdir, "-o", str(exe)])
    out = tmp_path / "result.bin"
    subprocess.check_call([str(exe), str(blob), str(out)])
    raw = np.fromfile(out, np.uint8)
    # python path on the same data
    env.reset()
    off = 0
    for t in range(steps):
        obs, rew, term, trunc, info = env.step(torch.from_numpy(actions[t]))
        r = raw[off: off + 8 * E].view(np.float64); off += 8 * E
        tm = raw[off: off + E]; off += E
        inf = raw[off: off + 8 * E * 16].view(np.float64).reshape(E, 16); off += 8 * E * 16
        assert np.array_equal(r, rew.cpu().numpy()), t
        assert np.array_equal(tm, term.cpu().numpy()), t
        assert np.array_equal(inf, info.cpu().numpy()), t
    assert np.array_equal(raw[off:].reshape(E, 4, 150, 150), obs.cpu().numpy())
    env.close()
