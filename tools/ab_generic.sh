#!/bin/bash
# tools/ab_generic.sh OUT "ENV=.. ENV=.. -- bench args" ...   each configuration twice, interleaved
OUT=$1; shift
for rep in 1 2; do
  for cfg in "$@"; do
    envs="${cfg%% -- *}"; args="${cfg#* -- }"
    echo -n "$cfg: " >> $OUT
    env $envs python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-strong $args 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'steady', round(d.get('steady_state',{}).get('value',0)))" >> $OUT
  done
done
