#!/bin/bash
# Three-way mixed launch sweep (run on the GPU box): BP_MIX=<solo> BP_MIX_PLAIN=<one env per wave> rest packed two to a wave in cost order.
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['physics_ms'],3), 'steady', round(d['steady_state']['value']))"; }
echo -n "default: "; run
for cfg in "$@"; do
  s=${cfg%%:*}; p=${cfg##*:}
  echo -n "solo $s plain $p adjacent: "; BP_MIX=$s BP_MIX_PLAIN=$p BP_MIX_PAIR=adjacent run
done
echo -n "default: "; run
