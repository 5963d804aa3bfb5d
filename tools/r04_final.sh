#!/bin/bash
# round-4 final measurement on one GPU box: profile (kernel trace + PMC passes incl. the LDS pass), default bench, the informational envs, launch time against batch
# size, the --gpus 2 rehearsal over gloo (two ranks on the one device).   tools/r04_final.sh BUILD_ID
BUILD=${1:-unknown}
REPO=$(pwd); OUT=$REPO/gpurun_out/r04_final; mkdir -p $OUT
bash tools/profile_gpu.sh r04_final $BUILD > $OUT/profile.log 2>&1 || exit 1
cp -r $REPO/gpurun_out/prof_r04_final/summary.txt $REPO/gpurun_out/prof_r04_final/pmc.json $REPO/gpurun_out/prof_r04_final/kernel_stats.csv $REPO/gpurun_out/prof_r04_final/bench_under_rocprof.json $OUT/ 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
for E in 1024 2048 3072 4096 6144 8192; do
  echo -n "E=$E: "; python bench.py --steps 30 --warmup 5 --envs-per-gpu $E --no-cpu-baseline --no-steady-state 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['physics_ms'],3))"
done > $OUT/launch_vs_envs.txt
(python bench.py --env maze --steps 10 --warmup 3 --no-cpu-baseline; python bench.py --env box --steps 10 --warmup 3 --no-cpu-baseline; python bench.py --env area --steps 10 --warmup 3 --no-cpu-baseline; python bench.py --config c5 --no-cpu-baseline) > $OUT/bench_other_envs.jsonl 2> $OUT/bench_other.err
BP_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-steady-state > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
echo done > $OUT/done
