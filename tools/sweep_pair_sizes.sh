run() { local label=$1; shift; local out=$(env "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['physics_ms'],3), d.get('invalid'))"); echo "$label $out"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-steady-state"
run "16k default(solo nosched)" BP_PAIR=0 $B --envs-per-gpu 16384
run "16k fixed pairs" BP_PAIR=1 BP_BENCH_IGNORE_CAPACITY=1 $B --envs-per-gpu 16384
run "16k sched40+pair2 loose" BP_SCHED=40 BP_PAIR=2 BP_PAIR_SOLO=0 BP_PP_ACT=30 BP_PP_WORK=100 BP_PP_RATE=1000 $B --envs-per-gpu 16384
run "16k sched100+pair2 loose" BP_SCHED=100 BP_PAIR=2 BP_PAIR_SOLO=0 BP_PP_ACT=30 BP_PP_WORK=100 BP_PP_RATE=1000 $B --envs-per-gpu 16384
run "16k sched40+pair2 defaults" BP_SCHED=40 BP_PAIR=2 $B --envs-per-gpu 16384
run "8k pair2 loose" BP_PAIR=2 BP_PAIR_SOLO=0 BP_PP_ACT=30 BP_PP_WORK=100 BP_PP_RATE=1000 $B --envs-per-gpu 8192
run "8k pair2 mid" BP_PAIR=2 BP_PAIR_SOLO=0 BP_PP_ACT=20 BP_PP_WORK=30 BP_PP_RATE=150 $B --envs-per-gpu 8192
run "8k pair2 defaults" BP_PAIR=2 $B --envs-per-gpu 8192
run "8k default" BP_PAIR=0 $B --envs-per-gpu 8192
run "c5 default" BP_PAIR=0 $B --config c5
run "c5 pair2" BP_PAIR=2 $B --config c5
run "c5 pair2 work=14" BP_PAIR=2 BP_PP_WORK=14 BP_PP_RATE=100 $B --config c5
run "2k default" BP_PAIR=0 $B --envs-per-gpu 2048
run "2k pair2" BP_PAIR=2 $B --envs-per-gpu 2048
