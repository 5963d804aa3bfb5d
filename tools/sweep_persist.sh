#!/bin/bash
# resident-wavefront scheduler: switches, same box: tools/sweep_persist.sh OUT "VAR=.. VAR=.." "VAR=.." ...   (each argument one configuration; run twice, interleaved)
OUT=$1; shift
run() { echo -n "$*: " >> $OUT; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-strong 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'steady', round(d.get('steady_state',{}).get('value',0)))" >> $OUT; }
for rep in 1 2; do for cfg in "$@"; do run $cfg; done; done
