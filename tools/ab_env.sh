#!/bin/bash
# Same-box A/B of library builds on a secondary env (no roofline object): tools/ab_env.sh box|area|maze steps lib1.so lib2.so ...
ENV=$1; STEPS=$2; shift 2
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "$(basename $lib): "
    BP_PROF=1 BP_PROF_LIB=$lib python bench.py --env $ENV --steps $STEPS --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))"
  done
done
