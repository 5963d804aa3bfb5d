#!/bin/bash
# Same-box comparison of library builds on the three informational envs: tools/ab_others.sh lib1.so lib2.so ...   (one pass, 12 steps after 3 warm-up steps)
for env in maze box area; do
  for lib in "$@"; do
    echo -n "$env $(basename $lib): "
    BP_PROF=1 BP_PROF_LIB=$lib python bench.py --env $env --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['physics_ms'],3))"
  done
done
