#!/bin/bash
# Same-box sweep of the pairing knobs (run on the GPU box): default scheduler against BP_PAIR=2 with different solo shares and leave limits.
B="python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-steady-state ${BENCH_ARGS:-}"
run() { # label, env assignments...
  local label=$1; shift
  local out=$(env "$@" $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['physics_ms'],3), d.get('invalid'))")
  echo "$label $out"
}
run "sched-default" BP_PAIR=0
for solo in ${SOLOS:-0 256 1024}; do
  for act in ${ACTS:-8 12 18}; do
    for work in ${WORKS:-8 16 40}; do
      for rate in ${RATES:-100}; do
        run "pair2 solo=$solo act=$act work=$work rate=$rate" BP_PAIR=2 BP_PAIR_SOLO=$solo BP_PP_ACT=$act BP_PP_WORK=$work BP_PP_RATE=$rate
      done
    done
  done
done
run "sched-default" BP_PAIR=0
