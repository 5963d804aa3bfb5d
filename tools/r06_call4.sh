#!/bin/bash
# round 6, GPU call 4: the recurrence shortcut of execute_robot_path -- parity (box-delivery / area-clearing suites + the on/off test), then the same-box A/B
OUT=gpurun_out/r06_c4; mkdir -p $OUT
python -m pytest tests/test_gpu_bd.py tests/test_gpu_ac.py -x -q > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
B=benchpush_amd/libbenchpush_hip_base.so; N=benchpush_amd/libbenchpush_hip.so
for rep in 1 2; do for lib in $B $N; do
  echo -n "$(basename $lib) box: " >> $OUT/ab_box.txt
  BP_PROF=1 BP_PROF_LIB=$lib python bench.py --env box --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'ms/step', round(d['ms_per_step'],2), d.get('straggler_env_steps',{}).get('ran_into_STEP_LIMIT'))" >> $OUT/ab_box.txt
  echo -n "$(basename $lib) area: " >> $OUT/ab_box.txt
  BP_PROF=1 BP_PROF_LIB=$lib python bench.py --env area --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'ms/step', round(d['ms_per_step'],2))" >> $OUT/ab_box.txt
done; done
cat $OUT/ab_box.txt
