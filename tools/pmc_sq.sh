#!/bin/bash
# tools/pmc_sq.sh E "COUNTERS..."  -> per-kernel mean of the listed SQ counters for k_physics_step
E=$1; shift
export TMPDIR=/tmp; REPO=$(pwd); cd /tmp; export PYTHONPATH=$REPO
rm -rf $REPO/gpurun_out/pmcx
rocprofv3 --kernel-trace --pmc $@ --output-format csv -d $REPO/gpurun_out/pmcx -- python3 $REPO/bench.py ${PMC_BENCH_ARGS} --steps 4 --warmup 8 --envs-per-gpu $E --no-cpu-baseline > /dev/null 2>&1
cd $REPO; python3 - <<'PY'
import csv, glob, collections
sq = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcx/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        sq[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
import os
for k in os.environ.get('PMC_KERNELS', 'k_physics_step').split(','):
    print(k, ' '.join('%s=%.4g' % (c, sorted(v)[-1] if os.environ.get('PMC_MAX') else sum(v[-4:])/len(v[-4:])) for c, v in sorted(sq[k].items())))
PY
