#!/bin/bash
# Same-box sweep of the longest-remaining-first scheduling mode (BP_SCHED_LRPT=1) against least-advanced-first (run on the GPU box).
run() { local label=$1; shift; local out=$(env "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['physics_ms'],3), round(d.get('steady_state',{}).get('value',0)), d.get('invalid'))"); echo "$label $out"; }
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-}"
run "least-advanced-first" BP_SCHED_LRPT=0 $B
for bw in ${BWS:-4500 9000 18000}; do for hy in ${HYS:-0 1 2}; do for ch in ${CHS:-40}; do for fl in ${FLS:-150}; do
run "lrpt bw=$bw hyst=$hy chunk=$ch floor=$fl" BP_SCHED_LRPT=1 BP_SCHED_BW=$bw BP_SCHED_HYST=$hy BP_SCHED=$ch BP_SCHED_FLOOR=$fl $B
done; done; done; done
run "least-advanced-first" BP_SCHED_LRPT=0 $B
