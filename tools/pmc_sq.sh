#!/bin/bash
# tools/pmc_sq.sh E "COUNTERS..."  -> per-kernel mean of the listed SQ counters for k_physics_step over the last four profiled launches
# (default: launches 26..29 of fresh episodes, i.e. episodes 26-29 steps old, close to the steady-state mix; PMC_BENCH_ARGS overrides)
E=$1; shift
export TMPDIR=/tmp; REPO=$(pwd); cd /tmp; export PYTHONPATH=$REPO
rm -rf $REPO/gpurun_out/pmcx
rocprofv3 --kernel-trace --pmc $@ --output-format csv -d $REPO/gpurun_out/pmcx -- python3 $REPO/bench.py --envs-per-gpu $E --no-cpu-baseline ${PMC_BENCH_ARGS:---steps 6 --warmup 24 --no-steady-state} > /dev/null 2>&1
cd $REPO; python3 - <<'PY'
import csv, glob, collections
sq = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcx/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('k_physics_step_sched') and int(r['Grid_Size']) <= 256 * 64: continue   # completion launches
        sq[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
import os
for k in os.environ.get('PMC_KERNELS', 'k_physics_step').split(','):
    print(k, ' '.join('%s=%.4g' % (c, sorted(v)[-1] if os.environ.get('PMC_MAX') else sum(v[-4:])/len(v[-4:])) for c, v in sorted(sq[k].items())))
PY
