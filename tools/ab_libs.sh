#!/bin/bash
# Same-box A/B of library builds on the default bench (or with extra bench arguments): tools/ab_libs.sh OUT "bench args" lib1.so lib2.so ...   (interleaved, twice)
OUT=$1; ARGS=$2; shift 2
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "$(basename $lib) $ARGS: " >> $OUT
    BP_PROF=1 BP_PROF_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-strong $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'steady', round(d.get('steady_state',{}).get('value',0)))" >> $OUT
  done
done
