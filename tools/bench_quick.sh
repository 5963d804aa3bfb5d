#!/bin/bash
# usage: tools/bench_quick.sh [envs...]  -> value ms/step physics_ms raster_ms
for E in "$@"; do
python bench.py --steps 10 --warmup 3 --envs-per-gpu $E --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('E=%d value=%.0f ms/step=%.2f physics_ms=%.2f raster_ms=%.2f' % (d['config']['envs_per_gpu'], d['value'], d['ms_per_step'], d['roofline']['physics_ms'], d['roofline']['raster_kernel']['ms']))"
done
