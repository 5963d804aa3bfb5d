#!/bin/bash
# Preemptive scheduler sweep (run on the GPU box): tools/sweep_sched.sh [bench args --] chunk1 chunk2 ...
ARGS=""
if [ "$1" = "--args" ]; then ARGS="$2"; shift 2; fi
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d.get('steady_state') or {}; print(round(d['value']), round(d['roofline']['physics_ms'],3), 'steady', round(s.get('value',0)), round(s.get('physics_ms',0),3))"; }
echo -n "default: "; BP_SCHED=0 run
for c in "$@"; do echo -n "BP_SCHED=$c: "; BP_SCHED=$c run; done
echo -n "default: "; BP_SCHED=0 run
