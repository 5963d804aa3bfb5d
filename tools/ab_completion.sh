for rep in 1 2; do
for c in 1 0; do
echo -n "completion=$c: "; BP_SCHED_COMPLETION=$c python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['physics_ms'],3))"
done; done
