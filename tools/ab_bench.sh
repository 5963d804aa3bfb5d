#!/bin/bash
# Same-box A/B of library builds: tools/ab_bench.sh [--env ship-ice] lib1.so lib2.so ...   (two alternating passes; devices differ by several
# per cent in wall time, so only numbers from one gpurun call are comparable)
ENV=ship-ice
if [ "$1" = "--env" ]; then ENV=$2; shift 2; fi
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "$(basename $lib): "
    BP_PROF=1 BP_PROF_LIB=$lib python bench.py ${AB_ARGS} --env $ENV --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['physics_ms'],3))"
  done
done
