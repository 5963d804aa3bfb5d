#!/bin/bash
# round-5 final measurement on one GPU box: profile (kernel trace + PMC passes incl. the LDS pass), default bench, the informational envs, launch time against batch
# size, the --gpus 2 rehearsal over gloo (two ranks on the one device).   tools/r05_final.sh BUILD_ID
BUILD=${1:-unknown}
REPO=$(pwd); OUT=$REPO/gpurun_out/r05_final; mkdir -p $OUT
bash tools/profile_gpu.sh r05_final $BUILD > $OUT/profile.log 2>&1 || exit 1
cp -r $REPO/gpurun_out/prof_r05_final/summary.txt $REPO/gpurun_out/prof_r05_final/pmc.json $REPO/gpurun_out/prof_r05_final/kernel_stats.csv $REPO/gpurun_out/prof_r05_final/bench_under_rocprof.json $OUT/ 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
# same box, pairing off (two environments per wavefront): the A/B of the round
BP_PAIR=0 python bench.py --no-cpu-baseline > $OUT/bench_pairing_off.json 2> $OUT/bench_pairing_off.err
# same box, the scheduler as it was before resident wavefronts and pace priorities (one workgroup per task from the hardware dispatcher, static priority classes)
BP_SCHED_PERSIST=0 BP_SCHED_DYNPRIO=0 python bench.py --no-cpu-baseline > $OUT/bench_dispatcher_driven.json 2> $OUT/bench_dispatcher_driven.err
BP_SCHED_PERSIST=1 BP_SCHED_DYNPRIO=0 python bench.py --no-cpu-baseline > $OUT/bench_resident_static_classes.json 2> $OUT/bench_resident_static_classes.err
for E in 1024 2048 3072 4096 5120 6144 7168 8192; do
  echo -n "E=$E: "; python bench.py --steps 30 --warmup 5 --envs-per-gpu $E --no-cpu-baseline --no-steady-state 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['physics_ms'],3))"
done > $OUT/launch_vs_envs.txt
for E in 2048 4096 5120 6144 8192 16384; do
  echo -n "pairing off E=$E: "; BP_PAIR=0 python bench.py --steps 30 --warmup 5 --envs-per-gpu $E --no-cpu-baseline --no-steady-state 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['physics_ms'],3))"
done >> $OUT/launch_vs_envs.txt
echo -n "E=16384: " >> $OUT/launch_vs_envs.txt; python bench.py --steps 20 --warmup 5 --envs-per-gpu 16384 --no-cpu-baseline --no-steady-state 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['physics_ms'],3))" >> $OUT/launch_vs_envs.txt
python tools/bench_vecenv.py > $OUT/bench_vecenv.jsonl 2> $OUT/bench_vecenv.err
# box-delivery: kernel timeline of the one-pass and the two-pass step (same deterministic steps)
mkdir -p $REPO/gpurun_out/r05_box
for b in 0 3000; do BP_BD_BUDGET=$b tools/kt_box_timeline.sh > $REPO/gpurun_out/r05_box/timeline_budget_$b.txt 2>&1; done
# the paired sub-step: counters of the default launch at 8192 envs (pairing on) -- instruction mix per env and sub-step
PMC_KERNELS=k_physics_step_schedr PMC_BENCH_ARGS="--steps 6 --warmup 24 --no-steady-state" tools/pmc_sq.sh 8192 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH > $OUT/pmc_paired_8192.txt 2>&1
PMC_KERNELS=k_physics_step_schedr PMC_BENCH_ARGS="--steps 6 --warmup 24 --no-steady-state" tools/pmc_sq.sh 8192 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU FETCH_SIZE >> $OUT/pmc_paired_8192.txt 2>&1
(python bench.py --env maze --steps 10 --warmup 3; python bench.py --env box --steps 10 --warmup 3; python bench.py --env area --steps 10 --warmup 3; python bench.py --config c5 --no-cpu-baseline) > $OUT/bench_other_envs.jsonl 2> $OUT/bench_other.err
# (two ranks share the ONE device here: resident kernels of two processes can only alternate by wave save / restore -- 80 ms per launch in one run -- so the rehearsal
# uses the dispatcher-driven kernels; on a real node every rank has its own GPU)
BP_SCHED_PERSIST=0 BP_PAIR_RESIDENT=0 BP_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-steady-state > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
echo done > $OUT/done
