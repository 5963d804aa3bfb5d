B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-steady-state --envs-per-gpu 8192"
run() { local label=$1; shift; local out=$(env "$@" $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['physics_ms'],3), d.get('invalid'))"); echo "$label $out"; }
run "8192 sched-default" BP_PAIR=0
run "8192 nosched" BP_SCHED=0
run "8192 fixed pairs" BP_PAIR=1 BP_BENCH_IGNORE_CAPACITY=1
run "8192 pair2 solo=0 act=12 work=16 yield" BP_PAIR=2 BP_PAIR_SOLO=0 BP_PP_NOYIELD=0
run "8192 pair2 solo=0 act=20 work=40 yield" BP_PAIR=2 BP_PAIR_SOLO=0 BP_PP_NOYIELD=0 BP_PP_ACT=20 BP_PP_WORK=40
run "8192 pair2 solo=0 act=30 work=100 yield" BP_PAIR=2 BP_PAIR_SOLO=0 BP_PP_NOYIELD=0 BP_PP_ACT=30 BP_PP_WORK=100
run "8192 pair2 solo=1024 act=20 work=40 yield" BP_PAIR=2 BP_PAIR_SOLO=1024 BP_PP_NOYIELD=0 BP_PP_ACT=20 BP_PP_WORK=40
run "8192 pair2 solo=0 act=20 work=40 noyield" BP_PAIR=2 BP_PAIR_SOLO=0 BP_PP_NOYIELD=1 BP_PP_ACT=20 BP_PP_WORK=40
run "8192 pair2 solo=0 act=30 work=100 noyield" BP_PAIR=2 BP_PAIR_SOLO=0 BP_PP_NOYIELD=1 BP_PP_ACT=30 BP_PP_WORK=100
