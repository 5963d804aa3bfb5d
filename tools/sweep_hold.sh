run() { local label=$1; shift; local out=$(env "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['physics_ms'],3), round(d.get('steady_state',{}).get('value',0)), d.get('invalid'))"); echo "$label $out"; }
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline"
run "nopair hold=0" BP_PAIR=0 $B
run "nopair hold=1" BP_PAIR=0 BP_SCHED_HOLD=1 $B
for solo in 512 1024; do for snake in 0 1; do
run "pair hold=1 solo=$solo snake=$snake act=16 work=9" BP_PAIR=2 BP_SCHED_HOLD=1 BP_PAIR_SOLO=$solo BP_PP_SNAKE=$snake $B
run "pair hold=1 solo=$solo snake=$snake act=20 work=20 rate=110" BP_PAIR=2 BP_SCHED_HOLD=1 BP_PAIR_SOLO=$solo BP_PP_SNAKE=$snake BP_PP_ACT=20 BP_PP_WORK=20 BP_PP_RATE=110 $B
done; done
run "pair hold=0 solo=512" BP_PAIR=2 BP_PAIR_SOLO=512 $B
run "nopair hold=1" BP_PAIR=0 BP_SCHED_HOLD=1 $B
run "nopair hold=0" BP_PAIR=0 $B
