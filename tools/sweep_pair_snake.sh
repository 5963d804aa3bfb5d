run() { local label=$1; shift; local out=$(env "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['physics_ms'],3), round(d.get('steady_state',{}).get('value',0)), d.get('invalid'))"); echo "$label $out"; }
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline"
run "nopair" BP_PAIR=0 $B
for snake in 0 1; do for ho in 0 1; do for act in 12 20; do for work in 9 24; do
run "solo=0 snake=$snake heavyonly=$ho act=$act work=$work" BP_PAIR_SOLO=0 BP_PP_SNAKE=$snake BP_PP_HEAVY_ONLY=$ho BP_PP_ACT=$act BP_PP_WORK=$work $B
done; done; done; done
run "nopair" BP_PAIR=0 $B
