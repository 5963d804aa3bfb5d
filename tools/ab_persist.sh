#!/bin/bash
# Same-box A/B of the scheduled launch with workgroups from the hardware dispatcher (BP_SCHED_PERSIST=0) against resident wavefronts (default, 1 = one workgroup
# per wave slot): tools/ab_persist.sh OUT [bench.py arguments, e.g. --envs-per-gpu 4096 | --env maze]
OUT=$1; shift
for rep in 1 2; do
  for v in 0 1; do
    echo -n "BP_SCHED_PERSIST=$v $*: " >> $OUT
    BP_SCHED_PERSIST=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-strong "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'steady', round(d.get('steady_state',{}).get('value',0)))" >> $OUT
  done
done
