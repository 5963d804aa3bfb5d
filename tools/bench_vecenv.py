#!/usr/bin/env python3
"""What is left of the env throughput behind the SB3-shaped adapter.

    python tools/bench_vecenv.py [--envs 4096] [--steps 40] [--warmup 5] [--policy cnn|resnet18|none] [--amp]

Four legs on the same trials and the same policy network (the SmallCnn of examples/rollout_cnn_policy.py: uint8 observations straight from the env kernels
into a torch CNN on the same GPU), one JSON line each:
  raw          BatchedShipIceEnv.step + masked reset, device tensors, no adapter
  vec_device   BatchedVecEnv(to_numpy=False): VecEnv protocol, tensors stay on the device, no host synchronisation per step, lazy infos
  vec_numpy    BatchedVecEnv(to_numpy=True): what SB3 itself consumes -- one non-blocking copy per output into pinned host buffers, one sync per step
  vec_numpy_infos  the same with every infos[i] materialised each step (what a per-env python consumer would pay)
The reference's learners sit behind this interface (baselines/ship_ice_nav/ppo/policy.py:29-69).

`--policy resnet18` puts the reference's real extractor in the loop: PPO("CnnPolicy") with the ResNet18 features extractor of
baselines/feature_extractors.py:11-45 (torchvision's resnet18 with a 4-channel first convolution, the classifier removed: 512 features) and a linear action
head -- written out below because torchvision is not in the image; random weights, eval mode, fp32 like SB3 (`--amp`: bf16 autocast).  Extra legs for it:
  policy_only      the network alone on a resident batch of observations (what the learner costs per rollout step)
  raw_two_groups   two half-batches on two streams, each `policy -> env.step` in turn: the only way a synchronous rollout can overlap the policy with the
                   env step (the action needs the observation, the step needs the action).  A resident step kernel holds every wave slot for its whole
                   launch, so beside it nothing overlaps; BP_SCHED_PERSIST=0 (one workgroup per task from the hardware dispatcher) lets the other group's
                   convolutions slip into freed slots.  The summary prints how much of the shorter of the two (policy, env) the pipeline hides."""
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


def make_resnet18(in_channels, action_dim):
    """torchvision.models.resnet18 restated (conv 7x7 / 2, max-pool, four stages of two BasicBlocks with 64 / 128 / 256 / 512 planes, global average pool)
    with the reference's changes: `in_channels` input planes, no classifier; plus the policy's action head."""
    import torch.nn as nn

    class Block(nn.Module):
        def __init__(self, cin, cout, stride):
            super().__init__()
            self.c1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False); self.b1 = nn.BatchNorm2d(cout)
            self.c2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False); self.b2 = nn.BatchNorm2d(cout)
            self.down = None
            if stride != 1 or cin != cout:
                self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

        def forward(self, x):
            y = torch.relu(self.b1(self.c1(x)))
            y = self.b2(self.c2(y))
            return torch.relu(y + (x if self.down is None else self.down(x)))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            layers = [nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1)]
            cin = 64
            for cout, stride in ((64, 1), (128, 2), (256, 2), (512, 2)):
                layers += [Block(cin, cout, stride), Block(cout, cout, 1)]
                cin = cout
            self.features = nn.Sequential(*layers, nn.AdaptiveAvgPool2d(1), nn.Flatten())
            self.head = nn.Sequential(nn.Linear(512, action_dim), nn.Tanh())

        def forward(self, obs_u8):
            return self.head(self.features(obs_u8.float() / 255.0))

    return Net()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--policy", default="cnn", choices=["cnn", "resnet18", "none"])
    ap.add_argument("--amp", action="store_true", help="bf16 autocast for the policy network")
    ap.add_argument("--policy-chunk", type=int, default=0, help="evaluate the policy in chunks of this many envs (0 = the whole batch at once)")
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
    policy = (make_resnet18(4, 1) if args.policy == "resnet18" else SmallCnn(4, False, 1)).to(dev).eval()
    if args.policy == "resnet18":
        policy = policy.to(memory_format=torch.channels_last)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    noise = (torch.rand((K + W, E), generator=g, device=dev, dtype=torch.float64) * 2 - 1)   # U(-1, 1) as in bench.py, added to the policy's mean

    def act_of(obs_dev, t):
        if args.policy == "none":
            return noise[t]
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.amp):
            if args.policy_chunk > 0:
                mean = torch.cat([policy(c) for c in obs_dev.split(args.policy_chunk)])
            else:
                mean = policy(obs_dev)
            n = noise[t] if noise.shape[1] == mean.shape[0] else noise[t, : mean.shape[0]]
            return (0.1 * mean.squeeze(-1).double() + n).clamp_(-1, 1)

    results = {}
    for leg in args.legs.split(","):
        if leg == "policy_only":
            obs = torch.randint(0, 256, (E, 4, 150, 150), dtype=torch.uint8, device=dev)
            for t in range(W):
                act_of(obs, t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(W, W + K):
                act_of(obs, t)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            results[leg] = {"leg": leg, "envs": E, "steps": K, "policy": args.policy, "amp": args.amp, "env_steps_per_s": E * K / dt, "ms_per_step": dt / K * 1e3}
            print(json.dumps(results[leg]), flush=True)
            continue
        if leg == "raw_two_groups":
            # two half-batches, each on its own stream: policy(A) -> step(A) while policy(B) -> step(B); the env of a group waits for its own actions only
            H = E // 2
            envs = [BatchedShipIceEnv(H, cfg={"concentration": 0.3}, trials=trials, device=dev, env_id_offset=k * H) for k in range(2)]
            streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
            obs2 = []
            for k in range(2):
                with torch.cuda.stream(streams[k]):
                    obs2.append(envs[k].reset()[0])

            def step2(t):
                for k in range(2):
                    with torch.cuda.stream(streams[k]):
                        a = act_of(obs2[k], t)
                        o, rew, term, trunc, info = envs[k].step(a)
                        envs[k].reset(term)
                        obs2[k] = o
            for t in range(W):
                step2(t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(W, W + K):
                step2(t)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            for e_ in envs:
                e_.check_errors()
                e_.close()
            results[leg] = {"leg": leg, "envs": E, "steps": K, "policy": args.policy, "amp": args.amp, "env_steps_per_s": E * K / dt, "ms_per_step": dt / K * 1e3,
                            "resident_workgroups": None, "BP_SCHED_PERSIST": os.environ.get("BP_SCHED_PERSIST")}
            print(json.dumps(results[leg]), flush=True)
            del envs
            torch.cuda.empty_cache()
            continue
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
        results[leg] = {"leg": leg, "envs": E, "steps": K, "policy": args.policy, "amp": args.amp, "env_steps_per_s": E * K / dt, "ms_per_step": dt / K * 1e3,
                        "resident_workgroups": int(env.L.bp_sched_resident(env.h)), "BP_SCHED_PERSIST": os.environ.get("BP_SCHED_PERSIST")}
        print(json.dumps(results[leg]), flush=True)
        env.close()
        del env
        torch.cuda.empty_cache()
    if "raw" in results and "policy_only" in results:
        # how much of the policy a rollout hides: raw = policy + env in sequence; the env alone is what bench.py measures (--policy none)
        pol = results["policy_only"]["ms_per_step"]
        seq = results["raw"]["ms_per_step"]
        line = {"policy_ms": pol, "sequential_ms": seq, "env_ms_by_difference": seq - pol}
        if "raw_two_groups" in results:
            two = results["raw_two_groups"]["ms_per_step"]
            line.update(two_groups_ms=two, hidden_ms=seq - two, hidden_share_of_the_shorter=(seq - two) / max(min(pol, seq - pol), 1e-9))
        print(json.dumps({"overlap": line, "BP_SCHED_PERSIST": os.environ.get("BP_SCHED_PERSIST"), "policy": args.policy, "amp": args.amp}))
    if "raw" in results:
        print(json.dumps({"summary": {k: round(v["env_steps_per_s"] / results["raw"]["env_steps_per_s"], 4) for k, v in results.items()},
                          "what": "throughput of each leg / raw (same box, same trials, same policy)"}))


if __name__ == "__main__":
    main()
