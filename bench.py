#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of batched ship-ice-v0 env.step() on MI355X (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one batched env.step() over all envs of a rank: 400 physics sub-steps + work/reward + the on-device episode
metrics + the 4x150x150 u8 observation raster, followed by the auto-reset of the envs that terminated.  What is timed for a
reset is a copy of the trial's settled template (each trial's 1000 settle sub-steps run once, at load: reset() is a pure
function of the trial when cfg.random_start is off -- tested identical to settling in place) plus its first observation.
Inputs (scenarios, actions) are resident in HBM before the timed region.  `value` is measured from freshly reset episodes
(steps W..W+K); `steady_state` repeats the measurement with the envs spread uniformly over the phases of an episode.
Multi-GPU: environments shard over ranks (4096 per GPU, weak scaling, no data-path collective); the only collective is the
all-gather of the [E/R, 6] episode-metric rows after the timed region (RCCL over xGMI).  For N > 1 the line also carries
`allgather_ms` (median of 20 timed gathers of the real block on device tensors) and `strong_scaling` (a second timed region with
the same 4096 envs split over the N ranks: value, ms_per_step, speed-up and efficiency against this run's own one-GPU time).

    --config c2 (default)  BASELINE.json configs[1]: 4096 envs per GPU, 30 % concentration
    --config c5            BASELINE.json configs[4]: 4096 envs per GPU, 50 % concentration (32 768 envs on 8 GPUs)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# dmabuf IPC is what RCCL needs on this pool (without it: hipIpcGetMemHandle: invalid argument).  Set before anything touches the GPU, so that a rank started
# by an external torchrun / torch.distributed.run gets it as well as the ranks spawn_ranks() starts.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes_per_env_step(nb, nf, maxv=20, obs_bytes=4 * 150 * 150):
    """SURVEY.md section 8(d) accounting, restated for the binary64 state this build keeps.

    A_min   : state resident on-chip for the 400 sub-steps (what this design does):
              body state read+write 2*nb*9*8 B + geometry read nf*(2*V+3)*8 B + observation write + scalars.
    A_stream: body state streamed from HBM every sub-step: 400*nb*9*8*2 B + contact cache + A_min's geometry/obs terms.
    """
    state = 2 * nb * 9 * 8
    geom = nf * (2 * maxv + 3) * 8
    scal = 256
    a_min = state + geom + obs_bytes + scal
    a_stream = 400 * nb * 9 * 8 * 2 + 400 * 64 * 64 + geom + obs_bytes + scal
    return a_min, a_stream, state + geom + scal


def effective_cores():
    """Host cores this process may actually use: the cgroup CPU quota when it is below the visible CPU count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:           # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = f.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) // int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(env, trials):
    """Oracle (CPU restatement, oracle/bp_oracle.c) timed on the host cores of this box, OpenMP over envs, on a bounded
    sample of the same workload: same trials, same 400 sub-steps x 10 iterations + observation raster per env.step()."""
    from benchpush_amd.scenario import pack_trials
    from oracle import oracle as orc

    cores = effective_cores()
    nenv, steps = 8 * cores, 30   # 30 steps from a fresh reset: the same episode phases as the GPU's timed region (steps 5..35)
    pk = pack_trials(trials[: min(len(trials), nenv)], max_verts=24)
    cfg = env.cfg
    orc.bench(env.params, cfg.ship.vertices, cfg.ship.head, cfg.ship.tail, pk, min(nenv, 4), 1, cores)  # warm the library
    n, sec = orc.bench(env.params, cfg.ship.vertices, cfg.ship.head, cfg.ship.tail, pk, nenv, steps, cores)
    return {"value": n / sec, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d env.step() (%.0f%% ship-ice trials, 400 sub-steps x 10 iterations + raster each), OpenMP on %d "
                      "threads, step loop only (resets untimed), %.2f s wall = %.0f core-seconds" % (nenv, steps, float(cfg.concentration) * 100, cores, sec, sec * cores)}


def cpu_baseline_single_thread(env):
    """SURVEY 8d / BASELINE.md 3 (config C1): ONE env, 10 % concentration, action 0, one host thread -- the latency leg of the CPU
    baseline (the reference's own loop is one env in one python process)."""
    from benchpush_amd.config import default_cfg, merge_user_cfg, ship_ice_physics_params
    from benchpush_amd.envs.ship_ice import default_trials
    from oracle.oracle import OracleShipIce

    cfg = merge_user_cfg(default_cfg("ship_ice"), {"concentration": 0.1})
    trial = default_trials(0.1, 1, base_seed=0)[0]
    o = OracleShipIce(ship_ice_physics_params(cfg), cfg.ship.vertices, cfg.ship.head, cfg.ship.tail)
    o.reset(trial)
    n, t0 = 0, time.perf_counter()
    while n < 300 and time.perf_counter() - t0 < 6.0:
        _, _, term, _ = o.step(0.0)
        n += 1
        if term:
            o.reset(trial)
    sec = time.perf_counter() - t0
    return {"value": n / sec, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "1 env, 10%% concentration (%d floes), action 0, %d env.step() incl. raster, one thread, %.2f s" % (len(trial["obstacles"]), n, sec)}


def profile_sourced(name):
    """Numbers that need hardware counters come from the committed rocprofv3 passes of the same kernels, tagged with their source."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(path))
    except Exception:
        return None


def cpu_baseline_box(env, trials):
    """box-delivery oracle (oracle/bp_oracle_bd.c) on one host core: a bounded sample of the same trials and action stream."""
    from benchpush_amd import box_delivery_scenario as S
    from oracle.oracle_bd import OracleBoxDelivery

    nenv, steps = 8, 12
    rng = np.random.RandomState(0)
    orcs = []
    for e in range(nenv):
        bp = dict(env.bd_params)
        o = OracleBoxDelivery(S.box_delivery_physics_params(env.cfg), bp, env.cfg)
        o.reset(trials[e % len(trials)], observe=False)
        orcs.append(o)
    t0 = time.perf_counter()
    for _ in range(steps):
        for o in orcs:
            o.step(float(rng.uniform(-1, 1)))
    sec = time.perf_counter() - t0
    return {"value": nenv * steps / sec, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "%d envs x %d env.step() of the same trials (heading actions U(-1,1), observation included), one thread, %.2f s" % (nenv, steps, sec)}


def cpu_baseline_area(env, trials):
    """area-clearing oracle on one host core: a bounded sample of the same trials with heading actions."""
    from benchpush_amd import area_clearing_scenario as A
    from oracle.oracle_bd import OracleAreaClearing

    nenv, steps = 8, 12
    rng = np.random.RandomState(0)
    orcs = []
    for e in range(nenv):
        o = OracleAreaClearing(A.area_clearing_physics_params(env.cfg), dict(env.bd_params), env.cfg)
        o.reset(trials[e % len(trials)], observe=False)
        orcs.append(o)
    t0 = time.perf_counter()
    for _ in range(steps):
        for o in orcs:
            o.step(float(rng.uniform(-1, 1)))
    sec = time.perf_counter() - t0
    return {"value": nenv * steps / sec, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "%d envs x %d env.step() of the same trials (heading actions U(-1,1), observation included), one thread, %.2f s" % (nenv, steps, sec)}


def cpu_baseline_maze(env, layouts):
    """maze-NAMO oracle (oracle/bp_oracle.c) on one host core: a bounded sample of the same layouts, U(-1, 1) actions, 400 sub-steps + rotated raster per step."""
    from oracle.oracle import OracleMaze

    nenv, steps = 8, 25
    rng = np.random.RandomState(0)
    c = env.cfg
    orcs = []
    for e in range(nenv):
        o = OracleMaze(env.params, c.robot.vertices, c.robot.wheel_vertices, c.obstacle_size)
        o.reset(layouts[e % len(layouts)])
        orcs.append(o)
    t0 = time.perf_counter()
    n = 0
    for _ in range(steps):
        for e, o in enumerate(orcs):
            _, _, term, _ = o.step(float(rng.uniform(-1, 1)))
            n += 1
            if term:
                o.reset(layouts[e % len(layouts)])
    sec = time.perf_counter() - t0
    return {"value": n / sec, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "%d envs x %d env.step() of the same layouts (20 boxes, U(-1,1) actions, observation included; resets inside the loop), one thread, %.2f s"
                      % (nenv, steps, sec)}


def spawn_ranks(n):
    """Start `n` ranks of this script (one per GPU) as a child `torch.distributed.run` on 127.0.0.1 and return its exit code.

    Nothing here touches the GPU (`torch.cuda.device_count()` does not initialise it on this image), the child is a subprocess and not an
    exec, and a box with fewer GPUs than ranks is refused unless BP_BENCH_BACKEND=gloo asks for the device-sharing plumbing check."""
    import socket
    import subprocess
    backend = os.environ.get("BP_BENCH_BACKEND", "nccl")
    plumbing = "--plumbing-only" in sys.argv
    if backend == "nccl" and not plumbing and torch.cuda.device_count() < n:
        print("bench.py: --gpus %d but %d GPUs visible; refusing (BP_BENCH_BACKEND=gloo runs the ranks on shared devices as a plumbing check)"
              % (n, torch.cuda.device_count()), file=sys.stderr)
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def time_allgather(rows, cnt, dist, sync, reps=20):
    """Median wall time in ms (this rank's own clock; rank 0's is printed) of `gather_episode_block` on the given tensors -- the one collective of the path ("RCCL all-gather time").  `sync` drains the
    device (torch.cuda.synchronize for nccl, a no-op for CPU tensors); a barrier before each repetition aligns the ranks."""
    from benchpush_amd.parallel import gather_episode_block
    ts = []
    for _ in range(reps):
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        gather_episode_block(rows, cnt, dist)
        sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return {"median_ms": ts[len(ts) // 2], "min_ms": ts[0], "max_ms": ts[-1], "reps": reps,
            "payload_bytes_per_rank": int(rows.shape[0]) * (int(rows.shape[1]) + 1) * 8,
            "what": "gather_episode_block: one all_gather_into_tensor of the [E/R, 7] float64 episode block (%s tensors, backend %s)"
                    % (rows.device.type, dist.get_backend() if dist is not None else "none")}


def plumbing_only(args, rank, world):
    """The N > 1 path without an environment: process group, barrier, the episode-block all-gather with the real [E/R, 6] shape.  Rank 0
    prints a line whose `n_gpus` is the number of ranks that joined the group (what the CPU test of `--gpus N` checks)."""
    import torch.distributed as dist
    from benchpush_amd.parallel import gather_episode_block, summarize_episode_block
    joined = 1
    allr = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("BP_BENCH_BACKEND", "gloo"))
        joined = dist.get_world_size()
        E = args.envs_per_gpu
        rows = torch.full((E, 6), float(rank), dtype=torch.float64)
        cnt = torch.ones(E, dtype=torch.int64)
        dist.barrier()
        allr, allc = gather_episode_block(rows, cnt, dist)
        summarize_episode_block(allr, allc)
        ag = time_allgather(rows, cnt, dist, lambda: None)
        # the per-rank block of the measured N > 1 line (a straggler rank shows in ms_per_step_by_rank): here every rank reports its all-gather median
        mine = torch.tensor([ag["median_ms"], 0.0], dtype=torch.float64)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank = [float(t_[0].item()) for t_ in allt]
        dist.barrier()
    if rank == 0:
        line = {"metric": "plumbing only: rank launch + rendezvous + episode-block all-gather", "value": None, "n_gpus": joined,
                "plumbing_only": True, "gathered_shape": None if allr is None else [int(allr.shape[0]), int(allr.shape[1])]}
        if world > 1:   # the keys the measured N > 1 line carries (no environment here: the strong-scaling block names its split only)
            line["allgather_ms"] = ag["median_ms"]
            line["allgather"] = ag
            line["ranks"] = {"ms_per_step_min": None, "ms_per_step_max": None, "ms_per_step_by_rank": [None] * world, "physics_ms_by_rank": [None] * world,
                             "slowest_rank": None, "allgather_median_ms_by_rank": per_rank, "note": "plumbing only: no environment was stepped"}
            line["env"] = {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
            line["strong_scaling"] = {"total_envs": args.envs_per_gpu, "envs_per_gpu": args.envs_per_gpu // world, "value": None, "ms_per_step": None,
                                      "speedup_vs_one_gpu": None, "efficiency": None, "note": "plumbing only: no environment was stepped"}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    return 0


def _check_errors(env):
    """Capacity / scheduler errors of the timed steps are fatal, except in kernel experiments that shrink capacities on purpose
    (BP_BENCH_IGNORE_CAPACITY=1: the error is printed to stderr and the line is marked)."""
    try:
        env.check_errors()
    except Exception as e:  # noqa: BLE001
        if os.environ.get("BP_BENCH_IGNORE_CAPACITY") != "1":
            raise
        print("bench.py: ignoring %s" % e, file=sys.stderr)
        global _IGNORED_ERRORS
        _IGNORED_ERRORS = True


_IGNORED_ERRORS = False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--concentration", type=float, default=0.3)
    ap.add_argument("--trials", type=int, default=100)
    ap.add_argument("--config", default="c2", choices=["c2", "c5"], help="c2 = BASELINE.json configs[1] (30 %%), c5 = configs[4] (50 %%, 4096 envs per GPU)")
    ap.add_argument("--no-steady-state", action="store_true", help="skip the staggered-phase (steady-state) measurement")
    ap.add_argument("--steady-steps", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling region (E envs split over the ranks)")
    ap.add_argument("--no-auto-reset", action="store_true")
    ap.add_argument("--env", default="ship-ice", choices=["ship-ice", "maze", "box", "area"],
                    help="ship-ice = BASELINE.json configs[1] (the headline); maze = configs[2], box = configs[3] (box-delivery-v0, "
                         "12 boxes), both informational")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="no environment and no GPU: only the rank launch, the rendezvous and the [E/R, 6] episode-block all-gather (CPU check of --gpus N)")
    args = ap.parse_args()
    if args.config == "c5":
        args.concentration = 0.5

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (a child torch.distributed.run, never an exec: nothing in
        # this process has touched the GPU yet) and leave with the child's exit code; rank 0 of the child prints the JSON line.
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if args.plumbing_only:
        return plumbing_only(args, rank, world)
    if world > 1 and os.environ.get("BP_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but %d GPUs visible (BP_BENCH_BACKEND=gloo shares devices for a plumbing check)"
                         % (world, torch.cuda.device_count()))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # BP_BENCH_BACKEND=gloo is a plumbing check of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices and the
        # two collectives run on CPU copies); the measured configuration is the default: one rank per GPU over RCCL
        backend = os.environ.get("BP_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            dist.init_process_group(backend)
    coll_device = torch.device("cuda", local_rank) if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
    from benchpush_amd.parallel import allgather_episode_metrics, gather_episode_block, summarize_episode_block

    E = args.envs_per_gpu
    if args.env == "maze":
        from benchpush_amd.envs.maze_namo import BatchedMazeEnv
        env = BatchedMazeEnv(E, cfg={"num_obstacles": 20}, num_layouts=args.trials, base_seed=0, device=device, env_id_offset=rank * E)
        trials = env.layouts
        nf_mean = 20.0
    elif args.env == "box":
        from benchpush_amd.envs.box_delivery import BatchedBoxDeliveryEnv
        env = BatchedBoxDeliveryEnv(E, cfg={"boxes": {"num_boxes_small": 12}}, num_trials=min(args.trials, 64), device=device,
                                    env_id_offset=rank * E)
        trials = env.trials
        nf_mean = 12.0
    elif args.env == "area":
        from benchpush_amd.envs.area_clearing import BatchedAreaClearingEnv
        env = BatchedAreaClearingEnv(E, num_trials=min(args.trials, 64), device=device, env_id_offset=rank * E)
        trials = env.trials
        nf_mean = 10.0
    else:
        trials = default_trials(args.concentration, args.trials, base_seed=0)
        env = BatchedShipIceEnv(E, cfg={"concentration": args.concentration}, trials=trials, device=device,
                                env_id_offset=rank * E)
        nf_mean = float(np.mean([len(t["obstacles"]) for t in trials]))
    K, W = args.steps, args.warmup
    # actions ~ U(-1,1), counter-style: a generator keyed by (seed, rank); resident in HBM before timing
    g = torch.Generator(device=device)
    g.manual_seed(1234 + rank)
    PH = 40                                                       # phases of the staggered (steady-state) measurement
    KS = 0 if (args.no_steady_state or args.env != "ship-ice") else args.steady_steps
    extra = (PH + KS) if KS > 0 else 0
    actions = (torch.rand((K + W + extra, E), generator=g, device=device, dtype=torch.float64) * 2 - 1).float().double()

    env.reset()
    ep_done = torch.zeros(E, dtype=torch.int64, device=device)
    ep_success = torch.zeros(E, dtype=torch.int64, device=device)

    def one_step(t):
        obs, rew, term, trunc, info = env.step(actions[t])
        if not args.no_auto_reset:
            ep_done.add_(term.to(torch.int64))
            if args.env in ("box", "area"):   # success = every box delivered / cleared (terminated without the truncation)
                done = (term | trunc) if args.env == "area" else term
                ep_done.add_((done.to(torch.int64) - term.to(torch.int64)))   # area-clearing also ends episodes by time truncation
                ep_success.add_((term.to(torch.int64) - (trunc.to(torch.int64) if args.env == "box" else 0)).clamp_min(0))
            else:
                ep_success.add_(info[:, 8].to(torch.int64))
            env.reset((term | trunc) if args.env == "area" else term)

    for t in range(W):
        one_step(t)
    torch.cuda.synchronize()
    clk0 = env.clock_stamps() if hasattr(env, "clock_stamps") and hasattr(env.L, "bp_get_clock_stamps") and args.env in ("ship-ice", "maze") else None
    env.enable_timing(True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(W, W + K):
        one_step(t)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    phys_ms, rast_ms, nlaunch = env.kernel_time_ms()
    cost_stats = env.cost_stats() if (args.env in ("ship-ice", "maze") and hasattr(env, "cost_stats") and hasattr(env.L, "bp_get_cost_stats")) else None
    env.enable_timing(False)
    _check_errors(env)
    # clock the chip held over the timed region: shader-clock counter against the 100 MHz reference, both stamped on the device after every step
    clk1 = env.clock_stamps() if clk0 is not None else None
    # both stamps of the pair come from ONE XCD (the shader-clock counters of different XCDs are not synchronised): the XCD with the longest span
    clock_hz, clock_xcd, clock_span_s = env.clock_hz_between(clk0, clk1, with_span=True) if clk1 is not None else (None, None, None)
    clock_xcds = env.clock_per_xcd(clk0, clk1) if (clk1 is not None and hasattr(env, "clock_per_xcd")) else None

    tmax = torch.tensor([dt], dtype=torch.float64, device=coll_device)
    rank_ms = None
    if dist is not None:
        # every rank's own wall time of the timed region (a straggler rank shows here; `value` uses the maximum)
        mine = torch.tensor([dt / K * 1e3, phys_ms], dtype=torch.float64, device=coll_device)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per = [float(t_[0].item()) for t_ in allt]
        rank_ms = {"ms_per_step_min": min(per), "ms_per_step_max": max(per), "ms_per_step_by_rank": per,
                   "physics_ms_by_rank": [float(t_[1].item()) for t_ in allt], "slowest_rank": int(np.argmax(per))}
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tmax = float(tmax.item())
    # episode metrics cross GPUs once, after the timed region (RCCL all-gather over xGMI): per-rank counters and, for ship-ice, the
    # [E/R, 6] rows of the on-device ShipIceMetric (efficiency, effort, reward, success, length, total_work per finished episode)
    local = torch.stack([ep_done.sum(), ep_success.sum()]).to(torch.float64).reshape(1, 2).to(coll_device)
    allm = allgather_episode_metrics(local, dist)
    episode_summary = None
    if args.env == "ship-ice":
        rows, cnt = env.episode_metrics()
        torch.cuda.synchronize()
        allr, allc = gather_episode_block(rows.to(coll_device), cnt.to(coll_device), dist)
        episode_summary = summarize_episode_block(allr, allc)
        episode_summary["gathered_shape"] = [int(allr.shape[0]), int(allr.shape[1])]
    total_envs = E * world
    value = total_envs * K / tmax
    allgather = None
    if dist is not None and args.env == "ship-ice":
        allgather = time_allgather(rows.to(coll_device), cnt.to(coll_device), dist,
                                   torch.cuda.synchronize if coll_device.type == "cuda" else (lambda: None))

    # ---- steady state: spread the envs uniformly over the phases of an episode (forced resets of E/PH envs per step, untimed), then
    #      time KS more steps; episodes run ~36 steps under U(-1,1) actions, so freshly reset batches are lighter than the mix ----
    steady = None
    if KS > 0:
        ids = torch.arange(E, device=device)
        for s_ in range(PH):
            one_step(W + K + s_)
            env.reset((ids % PH) == s_)
        torch.cuda.synchronize()
        env.kernel_time_ms()
        env.enable_timing(True)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for t in range(W + K + PH, W + K + PH + KS):
            one_step(t)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dts = time.perf_counter() - t1
        sp_ms, sr_ms, sn = env.kernel_time_ms()
        env.enable_timing(False)
        _check_errors(env)
        tm2 = torch.tensor([dts], dtype=torch.float64, device=coll_device)
        if dist is not None:
            dist.all_reduce(tm2, op=dist.ReduceOp.MAX)
        dts = float(tm2.item())
        steady = {"value": total_envs * KS / dts, "unit": "env-steps/s", "steps": KS, "ms_per_step": dts / KS * 1e3, "physics_ms": sp_ms,
                  "raster_ms": sr_ms, "phases": PH,
                  "how": "envs pre-advanced to uniformly spread episode steps (E/%d envs force-reset per step over %d untimed steps), "
                         "then %d timed steps with auto-reset" % (PH, PH, KS)}

    # ---- strong scaling: the SAME total of E envs split over the ranks, E / world per GPU, on the same ranks;
    #      T1 is this run's own weak region (every rank stepped E envs on one GPU), so speed-up = T1 / T_N and efficiency = speed-up / world ----
    strong = None
    if dist is not None and args.env == "ship-ice" and not args.no_strong and E % world == 0:
        Es = E // world
        env_s = BatchedShipIceEnv(Es, cfg={"concentration": args.concentration}, trials=trials, device=device, env_id_offset=rank * Es)
        env_s.reset()
        acts_s = actions[:, :Es].contiguous()

        def strong_step(t):
            _, _, term_s, _, _ = env_s.step(acts_s[t])
            if not args.no_auto_reset:
                env_s.reset(term_s)
        for t in range(W):
            strong_step(t)
        torch.cuda.synchronize()
        env_s.enable_timing(True)
        dist.barrier()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for t in range(W, W + K):
            strong_step(t)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dtS = time.perf_counter() - t2
        sp_ms2, _, _ = env_s.kernel_time_ms()
        env_s.enable_timing(False)
        _check_errors(env_s)
        tm3 = torch.tensor([dtS], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tm3, op=dist.ReduceOp.MAX)
        dtS = float(tm3.item())
        strong = {"total_envs": E, "envs_per_gpu": Es, "value": E * K / dtS, "unit": "env-steps/s", "steps": K, "ms_per_step": dtS / K * 1e3,
                  "physics_ms": sp_ms2, "one_gpu_ms_per_step": tmax / K * 1e3, "speedup_vs_one_gpu": tmax / dtS, "efficiency": tmax / dtS / world,
                  "how": "a second handle of E / world = %d envs per rank (global env ids rank * %d ..., same trials and action stream), same W + K steps, "
                         "barrier + synchronize on both sides, max over ranks; T1 = this run's weak region (E envs on every GPU)" % (Es, Es),
                  "expectation": "below ~3300 envs per GPU a launch lasts as long as its heaviest env's own chain (profiles/r05_final/launch_vs_envs.txt: "
                                 "12.5 ms at 1024 envs, 13.4 at 2048, 15.2 at 4096), so fixed E = 4096 over N GPUs gains at most ~1.2 x"}
        env_s.close()

    pairing = None
    if rank == 0 and args.env == "ship-ice" and hasattr(env, "pair_stats") and hasattr(env.L, "bp_get_pair_stats"):
        ps = env.pair_stats()
        if ps["mode"]:
            pairing = dict(ps, what="two environments per wavefront (lanes 0..31 / 32..63, one instruction stream; DESIGN.md section 4p): light envs advance in pairs, "
                                    "heavy ones alone; counters are cumulative since load (warm-up, timed and steady-state steps); BP_PAIR=0 turns it off")
    if rank == 0:
        sched_chunk = int(env.L.bp_sched_chunk(env.h)) if hasattr(env, "L") and hasattr(env.L, "bp_sched_chunk") else 0
        resident = int(env.L.bp_sched_resident(env.h)) if hasattr(env, "L") and hasattr(env.L, "bp_sched_resident") else 0
        nb = int(round(nf_mean)) + (11 if args.env == "maze" else 19 if args.env == "box" else 1)
        a_min, a_stream, a_phys = algorithmic_bytes_per_env_step(nb, nb - 1, maxv=20, obs_bytes=int(np.prod(env.obs_shape)))
        # SQ instruction mix / HBM traffic per launch need hardware counters: taken from the committed rocprofv3 passes of this kernel
        # (profiles/r06_final/pmc.json -- the newest earlier round's until this round's profile exists --, written by tools/profile_gpu.sh for the build named inside; config c2 only), never measured in this run
        pmc_dir = next((d for d in ("r06_final", "r05_final", "r04_final") if os.path.exists(os.path.join(ROOT, "profiles", d, "pmc.json"))), "r03_final")
        pmc = profile_sourced(os.path.join(pmc_dir, "pmc.json")) if (args.env == "ship-ice" and args.config == "c2") else None
        roof = {
            "bound": "issue",
            "accounting": "achieved / peak / frac are the HBM accounting of SURVEY 8d (algorithmic bytes per launch / kernel time against 8 TB/s); what binds "
                          "the kernel is named in 'bound' / 'binds', and 'issue' gives its share of the chip's wave-instruction issue slots",
            "clock_mhz": (clock_hz / 1e6) if clock_hz else None,
            "clock_source": ("s_memtime / s_memrealtime stamps of XCD %d around the timed launches" % clock_xcd) if clock_hz else
                            "no stamp pair from one XCD: issue.frac uses the nominal 2.1 GHz",
            "clock_span_ms": (clock_span_s * 1e3) if clock_span_s else None,
            "clock_per_xcd": clock_xcds,
            "clock_note": "shader-clock counter (s_memtime) over the 100 MHz reference counter (s_memrealtime) between the newest stamp before the timed region "
                          "and the newest after it, on one XCD; the span is printed because a reading over a few launches is noisier than one over the whole "
                          "region.  The guide's 2 400 MHz is the spec'd peak engine clock; this kernel draws little power (27 % of the lanes, no MFMA), and readings "
                          "of 2.37-2.62 GHz have been seen on different boxes of the pool -- the counter ratio is what the chip ran at, not a nominal value",
            "kernel": (("k_physics_step_sched" + (("r" if resident > 0 else "p") if pairing and pairing.get("mode") == 2 else "l" if resident > 0 else "") + ("_maze" if args.env == "maze" else ""))
                       if sched_chunk > 0 else ("k_physics_step_maze" if args.env == "maze" else "k_physics_step")),
            "pairing": pairing,
            "scheduler": ({"chunk_substeps": sched_chunk, "resident_workgroups": resident,
                           "what": "preemptive: envs parked at chunk boundaries while another is further behind, "
                           "least-advanced waiting env first (DESIGN.md 4a); resident_workgroups > 0: one workgroup per wave slot takes task after task itself "
                           "(k_physics_step_schedl; k_physics_step_schedr in pairing launches) instead of one workgroup per task from the hardware dispatcher, BP_SCHED_PERSIST=0 / BP_PAIR_RESIDENT=0 turn that off; "
                           "BP_SCHED=0 selects one wavefront per env for the whole step"}
                          if sched_chunk > 0 else None),
            "achieved": a_phys * E / (phys_ms * 1e-3) / 1e9 if phys_ms > 0 else None,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
            "binds": "wave issue slots and dependent-instruction latency of one wavefront per env, not HBM and not MFMA: the state stays "
                     "on-chip for the 400 sub-steps; with the preemptive scheduler the launch is within a few per cent of BOTH the chain of its "
                     "heaviest env and the sum of all chains / 2048 wave slots (DESIGN.md section 4a; both are measured in this run: roofline.ceiling).  Of the two the "
                     "chains bind: exact work reductions that reach only the light envs (two envs per wavefront, skipping the tail of the ~30 % of sub-steps without a "
                     "warm arbiter) left the launch where it was (DESIGN.md section 4d6)",
            "raster_kernel": {"kernel": {"ship-ice": "k_observe", "maze": "k_observe_maze"}.get(args.env, "k_bd_observe"),
                              "bytes_written_per_env": int(np.prod(env.obs_shape)),
                              "achieved": int(np.prod(env.obs_shape)) * E / (rast_ms * 1e-3) / 1e9 if rast_ms > 0 else None,
                              "ms": rast_ms, "frac": (int(np.prod(env.obs_shape)) * E / (rast_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if rast_ms > 0 else None},
            "physics_ms": phys_ms, "launches": nlaunch,
            "bytes_per_env_step": {"A_min": a_min, "A_stream": a_stream, "k_physics_min": a_phys},
            "hypothetical_stream_design": {"GB/s": (a_stream - 4 * 150 * 150) * E / (phys_ms * 1e-3) / 1e9 if phys_ms > 0 else None,
                                           "note": "bytes a design that streams the body state from HBM every sub-step would move, divided by "
                                                   "THIS kernel's time; nothing achieves it (SURVEY 8d asks for both accountings)"},
        }
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS if roof["achieved"] else None
        if pairing is not None and pairing.get("mode"):
            roof["ceiling"] = None
            roof["ceiling_note"] = ("pairing launch: an env's recorded cost is the time of the wavefront it shared (its mate's sub-steps included), so the sum over the envs is not "
                                    "the work of the launch; the two ceilings are printed for launches with one env per wavefront")
        elif cost_stats is not None and len(cost_stats) and phys_ms > 0:
            # the two lower bounds of a launch, from this run's own per-env cycle counts (bp_get_cost_stats) and this run's own clock
            ck_ = clock_hz if clock_hz else 2.4e9
            slots_ = int(env.L.bp_sched_resident(env.h)) if hasattr(env.L, "bp_sched_resident") else 0
            slots_ = slots_ if slots_ > 0 else 2048
            cs = cost_stats.astype(np.float64)
            work_ms = float(np.mean(cs[:, 0])) / slots_ / ck_ * 1e3
            chain_ms = float(np.mean(cs[:, 1])) / ck_ * 1e3
            roof["ceiling"] = {
                "heaviest_chain_ms": chain_ms, "work_over_slots_ms": work_ms,
                "launch_over_chain": phys_ms / chain_ms if chain_ms > 0 else None, "launch_over_work": phys_ms / work_ms if work_ms > 0 else None,
                "wave_slots": slots_, "launches": int(len(cost_stats)), "clock_used_mhz": ck_ / 1e6,
                "heaviest_chain_ms_max": float(np.max(cs[:, 1])) / ck_ * 1e3,
                "mean_env_busy_ms": float(np.mean(cs[:, 0])) / E / ck_ * 1e3,
                "what": "per timed launch, from the wave cycles every env's step took (its tasks summed, queue waits excluded; k_cost_stats after each launch): "
                        "work_over_slots = sum over the envs / wave slots of the device -- the launch if the slots were perfectly packed; heaviest_chain = the "
                        "largest of them -- the busy time of the heaviest env as it ran (beside a second wave on its SIMD; alone it is ~1.2 x shorter, DESIGN.md 4s). "
                        "Means over the timed launches, converted with the clock measured in this run; launch_over_* = physics_ms over each bound"}
        if pmc and phys_ms > 0:
            per = pmc.get("per_launch", {})
            insts = sum(per.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
            scale = E / float(pmc.get("envs", E))
            roof["traffic"] = (pmc.get("hbm_bytes_per_launch") * scale) if pmc.get("hbm_bytes_per_launch") else None   # per launch of E envs
            roof["wasted_traffic"] = (roof["traffic"] / (a_phys * E)) if roof["traffic"] else None
            roof["traffic_source"] = "profiles/%s/pmc.json (rocprofv3 --pmc passes of build %s, %s envs; not measured in this run)" % (
                pmc_dir, pmc.get("build", "?"), pmc.get("envs", "?"))
            if pmc.get("lds"):   # LDS bank conflicts beside occupancy and VALU utilisation (same profile-sourced passes)
                roof["lds_bank_conflict_frac"] = pmc["lds"].get("bank_conflict_frac")
                roof["lds"] = dict(pmc["lds"], source=roof["traffic_source"],
                                   what="bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra LDS-array cycles over all LDS-array cycles); "
                                        "issue_stall_share_of_wave_time = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES; array_busy_frac = LDS-array cycles / (256 CUs x kernel cycles)")
            ck = clock_hz if clock_hz else 2.1e9
            insts += per.get("SQ_INSTS_BRANCH", 0.0)
            roof["issue"] = {
                "wave_instructions_per_launch": insts * scale,
                "wave_instructions_per_env_substep": insts * scale / (E * env.params["steps"]),
                "frac": insts * scale / (1024 * ck * phys_ms * 1e-3),
                "clock_used_mhz": ck / 1e6,
                "unit": "wave-instructions per SIMD-cycle (1024 SIMDs x the measured clock x kernel time; VALU+SALU+branch+LDS+VMEM)",
                "valu_frac": per.get("SQ_INSTS_VALU", 0.0) * scale * 4 / (1024 * ck * phys_ms * 1e-3),
                "lanes_active": pmc.get("lanes_active"),
                "source": roof["traffic_source"],
            }
        out = {
            "metric": "env-steps/sec at N=4096 envs (ship-ice-v0), 1/2/4/8 MI355X" if args.env == "ship-ice"
                      else "env-steps/sec at N=4096 envs (maze-NAMO-v0), informational" if args.env == "maze"
                      else "env-steps/sec at N=4096 envs (box-delivery-v0), informational" if args.env == "box"
                      else "env-steps/sec at N=4096 envs (area-clearing-v0), informational",
            "value": value, "unit": "env-steps/s", "n_gpus": (dist.get_world_size() if dist is not None else 1), "steps": K, "warmup": W,
            "ms_per_step": tmax / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("ship-ice-v0, %d envs per GPU, %.0f%% concentration (mean %.1f floes), 400 sub-steps x 10 "
                                    "solver iterations per env.step, 4x150x150 u8 obs, auto-reset" % (E, args.concentration * 100, nf_mean))
                                   if args.env == "ship-ice" else
                                   ("maze-NAMO-v0, %d envs per GPU, 20 boxes, 400 sub-steps x 10 solver iterations per env.step, "
                                    "4x192x192 u8 rotated obs, auto-reset" % E) if args.env == "maze" else
                                   ("box-delivery-v0 (small_empty), %d envs per GPU, 12 boxes, heading actions, a variable number of "
                                    "2 ms sim steps per env.step (about 1000), one spfa per box + robot map, 224x224x4 u8 obs, auto-reset" % E)
                                   if args.env == "box" else
                                   ("area-clearing-v0 (clear_env), %d envs per GPU, 10 boxes, heading actions, path execution + 100 sim steps "
                                    "per env.step (about 950), robot spfa map, 224x224x4 u8 obs, auto-reset" % E),
                       "envs_per_gpu": E, "total_envs": total_envs, "concentration": args.concentration,
                       "substeps_per_step": env.params["steps"], "auto_reset": not args.no_auto_reset,
                       "episodes_finished": int(allm[:, 0].sum().item()), "episodes_success": int(allm[:, 1].sum().item()),
                       "config": args.config if args.env == "ship-ice" else None},
            "substeps_per_s": value * env.params["steps"],
            "roofline": roof,
        }
        if steady is not None:   # both accountings first-class: `value` = fresh episodes (steps W..W+K of a reset batch), `steady_state_value` = envs spread over all episode phases
            out["steady_state_value"] = steady["value"]
            out["value_note"] = ("value: the K timed steps right after reset + W warm-up steps (the contract's region); steady_state_value: the same batch after its envs were spread "
                                 "uniformly over the phases of an episode -- what a long-running job sees.  Quote the steady-state number for sustained throughput")
        if rank_ms is not None:
            out["ranks"] = rank_ms
        if allgather is not None:
            out["allgather_ms"] = allgather["median_ms"]
            out["allgather"] = allgather
        if dist is not None and args.env == "ship-ice":
            out["strong_scaling"] = strong
        if steady is not None:
            out["steady_state"] = steady
        if episode_summary is not None:
            out["episode_metrics"] = episode_summary
        if args.env in ("box", "area"):
            if hasattr(env, "stragglers") and hasattr(env.L, "bp_bd_get_stragglers"):
                resumed, limited = env.stragglers()
                bd_budget = int(env.L.bp_bd_budget(env.h)) if hasattr(env.L, "bp_bd_budget") else int(os.environ.get("BP_BD_BUDGET", "3000"))   # what the handle uses
                skips = env.cycle_skips() if hasattr(env.L, "bp_bd_get_cycle_skips") else (None, None)
                out["straggler_env_steps"] = {"ran_into_STEP_LIMIT": limited, "recurrences_found_in_execute_robot_path": skips[0], "sim_steps_skipped_by_them": skips[1], "finished_by_the_second_pass": resumed, "env_steps_total": E * (K + W),
                                              "budget_sim_steps": bd_budget,
                                              "what": "cumulative over warm-up and timed steps: env steps whose execute_robot_path / step_simulation_until_still loop hit the "
                                                      "reference's STEP_LIMIT (10 000 sim steps: single wavefronts that set the launch time), and env steps that ran past the "
                                                      "sim-step budget of the first pass and were finished by the second one beside the other envs' finish / map / observation kernels"}
            if (int(env.L.bp_bd_budget(env.h)) if hasattr(env.L, "bp_bd_budget") else int(os.environ.get("BP_BD_BUDGET", "3000"))) > 0:   # two-pass step: both groups' observation kernels run inside the physics window (other streams)
                out["roofline"]["raster_kernel"].update(ms=None, achieved=None, frac=None,
                                                        note="two-pass step (DESIGN.md 4c): k_bd_observe runs once per group on two streams inside physics_ms; "
                                                             "3.8-3.9 ms for 4 096 envs in the kernel trace (profiles/r05_box/)")
            out["roofline"]["kernel"] = "k_bd_physics (+ k_bd_plan / k_bd_finish / k_bd_robot_map / k_bd_observe in physics_ms)"
            out["roofline"]["note"] = ("persistent per-env wavefront over ~1000 sim steps; latency-bound like k_physics_step (DESIGN.md 4c).  A launch lasts as long as its slowest env: "
                                       "an env step whose step_simulation_until_still loop runs the reference's full 10 001 sim steps (box_delivery_env.py:990-1023; boxes that keep "
                                       "jittering against a wall, 8-9e8 wave cycles = 350-400 ms for ONE wavefront, no exact shortcut: the states never recur) sets 4 of 13 launches of "
                                       "this workload and two thirds of its time -- that is the reference's own loop, not kernel overhead.  The other kind -- execute_robot_path into "
                                       "STEP_LIMIT with a robot that pushes against a wall -- recurs bit for bit and is skipped exactly (straggler_env_steps.sim_steps_skipped_by_them)")
            out["substeps_per_s"] = None
        if world == 1 and not args.no_cpu_baseline and args.env == "ship-ice":
            out["cpu_baseline"] = cpu_baseline(env, trials)
            out["cpu_baseline"]["single_thread"] = cpu_baseline_single_thread(env)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        if world == 1 and not args.no_cpu_baseline and args.env in ("box", "area", "maze"):
            out["cpu_baseline"] = {"box": cpu_baseline_box, "area": cpu_baseline_area, "maze": cpu_baseline_maze}[args.env](env, trials)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        if _IGNORED_ERRORS:
            out["invalid"] = "capacity errors ignored (BP_BENCH_IGNORE_CAPACITY=1): experiment line, not a result"
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
