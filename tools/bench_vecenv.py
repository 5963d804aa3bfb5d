#!/usr/bin/env python3
"""What is left of the env throughput behind the SB3-shaped adapter.

    python tools/bench_vecenv.py [--envs 4096] [--steps 40] [--warmup 5] [--policy cnn|none]

Four legs on the same trials and the same policy network (the SmallCnn of examples/rollout_cnn_policy.py: uint8 observations straight from the env kernels
into a torch CNN on the same GPU), one JSON line each:
  raw          BatchedShipIceEnv.step + masked reset, device tensors, no adapter
  vec_device   BatchedVecEnv(to_numpy=False): VecEnv protocol, tensors stay on the device, no host synchronisation per step, lazy infos
  vec_numpy    BatchedVecEnv(to_numpy=True): what SB3 itself consumes -- one non-blocking copy per output into pinned host buffers, one sync per step
  vec_numpy_infos  the same with every infos[i] materialised each step (what a per-env python consumer would pay)
The reference's learners sit behind this interface (baselines/ship_ice_nav/ppo/policy.py:29-69)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--policy", default="cnn", choices=["cnn", "none"])
    ap.add_argument("--legs", default="raw,vec_device,vec_numpy,vec_numpy_infos")
    args = ap.parse_args()
    from rollout_cnn_policy import SmallCnn
    from benchpush_amd import _lib
    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
    from benchpush_amd.envs.vec_env import BatchedVecEnv
    E, K, W = args.envs, args.steps, args.warmup
    dev = torch.device("cuda", 0)
    trials = default_trials(0.3, 100, base_seed=0)
    torch.manual_seed(0)
    policy = SmallCnn(4, False, 1).to(dev).eval()
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    noise = (torch.rand((K + W, E), generator=g, device=dev, dtype=torch.float64) * 2 - 1)   # U(-1, 1) as in bench.py, added to the policy's mean

    def act_of(obs_dev, t):
        if args.policy == "none":
            return noise[t]
        with torch.no_grad():
            return (0.1 * policy(obs_dev).squeeze(-1).double() + noise[t]).clamp_(-1, 1)

    results = {}
    for leg in args.legs.split(","):
        env = BatchedShipIceEnv(E, cfg={"concentration": 0.3}, trials=trials, device=dev)
        if leg == "raw":
            obs, _ = env.reset()

            def step(t, obs):
                o, rew, term, trunc, info = env.step(act_of(obs, t))
                env.reset(term)
                return o
        else:
            venv = BatchedVecEnv(env, _lib.INFO_KEYS, max_episode_steps=300, to_numpy=(leg != "vec_device"))
            obs = venv.reset()

            def step(t, obs):
                od = obs if torch.is_tensor(obs) else torch.from_numpy(obs).to(dev, non_blocking=True)   # numpy path: the learner's own upload
                a = act_of(od, t)
                o, rew, done, infos = venv.step(a if leg == "vec_device" else a.float().cpu().numpy())
                if leg == "vec_numpy_infos":
                    n = 0
                    for i in range(len(infos)):
                        n += len(infos[i])
                return o
        for t in range(W):
            obs = step(t, obs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(W, W + K):
            obs = step(t, obs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        env.check_errors()
        results[leg] = {"leg": leg, "envs": E, "steps": K, "policy": args.policy, "env_steps_per_s": E * K / dt, "ms_per_step": dt / K * 1e3}
        print(json.dumps(results[leg]), flush=True)
        env.close()
        del env
        torch.cuda.empty_cache()
    if "raw" in results:
        print(json.dumps({"summary": {k: round(v["env_steps_per_s"] / results["raw"]["env_steps_per_s"], 4) for k, v in results.items()},
                          "what": "throughput of each leg / raw (same box, same trials, same policy)"}))


if __name__ == "__main__":
    main()
