run() { local label=$1; shift; local out=$(env "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['physics_ms'],3), round(d.get('steady_state',{}).get('value',0)), d.get('invalid'))"); echo "$label $out"; }
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline"
run "c2 pair default" BP_X=0 $B
for ch in 25 34 50 67; do run "c2 pair chunk=$ch" BP_SCHED=$ch $B; done
run "c2 nopair" BP_PAIR=0 $B
run "c5 pair default" BP_X=0 $B --config c5
run "c5 nopair" BP_PAIR=0 $B --config c5
run "c5 pair work=14 rate=100" BP_PP_WORK=14 BP_PP_RATE=100 $B --config c5
run "c5 pair solo=1024" BP_PAIR_SOLO=1024 $B --config c5
run "c2 pair solo=768 rate=100" BP_PAIR_SOLO=768 BP_PP_RATE=100 $B
run "c2 pair rate=50" BP_PP_RATE=50 $B
run "c2 pair keys=20 mv=24" BP_PP_KEYS=20 BP_PP_MV=24 $B
