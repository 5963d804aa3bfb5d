#!/bin/bash
# round-6 final measurement on one GPU box: profile (kernel trace + the PMC passes incl. TCC hit / miss), default bench, c5 with its CPU baseline, launch time against batch size,
# the informational envs, the box-delivery recurrence shortcut on / off, the VecEnv legs with the SmallCnn and with a ResNet18-sized policy (resident and dispatcher-driven),
# the --gpus 2 rehearsal over gloo (two ranks on the one device: exercises the shared-device fallback).   tools/r06_final.sh BUILD_ID
BUILD=${1:-unknown}
REPO=$(pwd); OUT=$REPO/gpurun_out/r06_final; mkdir -p $OUT
bash tools/profile_gpu.sh r06_final $BUILD > $OUT/profile.log 2>&1 || exit 1
cp -r $REPO/gpurun_out/prof_r06_final/summary.txt $REPO/gpurun_out/prof_r06_final/pmc.json $REPO/gpurun_out/prof_r06_final/kernel_stats.csv $REPO/gpurun_out/prof_r06_final/bench_under_rocprof.json $OUT/ 2>/dev/null
PMC_KERNELS=k_physics_step_schedl bash tools/pmc_sq.sh 4096 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum > $OUT/pmc_tcc.txt 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
python bench.py --config c5 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
for E in 1024 2048 3072 4096 5120 6144 7168 8192; do
  echo -n "E=$E: "; python bench.py --steps 30 --warmup 5 --envs-per-gpu $E --no-cpu-baseline --no-steady-state 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['roofline'].get('ceiling') or {}; print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['physics_ms'],3), 'chain', round(c.get('heaviest_chain_ms',0),2), 'work/slots', round(c.get('work_over_slots_ms',0),2))"
done > $OUT/launch_vs_envs.txt
echo -n "E=16384: " >> $OUT/launch_vs_envs.txt; python bench.py --steps 20 --warmup 5 --envs-per-gpu 16384 --no-cpu-baseline --no-steady-state 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['physics_ms'],3))" >> $OUT/launch_vs_envs.txt
(python bench.py --env maze --steps 10 --warmup 3; python bench.py --env box --steps 10 --warmup 3; python bench.py --env area --steps 10 --warmup 3) > $OUT/bench_other_envs.jsonl 2> $OUT/bench_other.err
# box-delivery / area-clearing: the recurrence shortcut of execute_robot_path on / off, interleaved three times
for rep in 1 2 3; do for cyc in 1 0; do for e in box area; do
  echo -n "$e BP_BD_CYCLE=$cyc: "; BP_BD_CYCLE=$cyc python bench.py --env $e --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['straggler_env_steps']['sim_steps_skipped_by_them'])"
done; done; done > $OUT/bd_recurrence_on_off.txt
python tools/bd_straggler_steps.py > $OUT/bd_straggler_steps.txt 2>&1
# VecEnv: SmallCnn legs as in round 5; then the reference's ResNet18-sized extractor, resident and dispatcher-driven step kernels
python tools/bench_vecenv.py > $OUT/bench_vecenv.jsonl 2> $OUT/bench_vecenv.err
python tools/bench_vecenv.py --policy resnet18 --steps 12 --warmup 3 --legs policy_only,raw,vec_device,raw_two_groups > $OUT/bench_vecenv_resnet18.jsonl 2> $OUT/bench_vecenv_resnet18.err
BP_SCHED_PERSIST=0 python tools/bench_vecenv.py --policy resnet18 --steps 12 --warmup 3 --legs policy_only,raw,raw_two_groups > $OUT/bench_vecenv_resnet18_dispatcher.jsonl 2>> $OUT/bench_vecenv_resnet18.err
python tools/bench_vecenv.py --policy resnet18 --amp --steps 12 --warmup 3 --legs policy_only,raw,raw_two_groups > $OUT/bench_vecenv_resnet18_bf16.jsonl 2>> $OUT/bench_vecenv_resnet18.err
# two ranks on the ONE device: the handles detect each other (residency lock) and launch the dispatcher-driven kernels by themselves
BP_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-steady-state > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
echo done > $OUT/done
