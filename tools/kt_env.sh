#!/bin/bash
# kernel-trace of a secondary bench: tools/kt_env.sh maze|box|area [steps]   (run on the GPU box via gpurun)
ENV=${1:-maze}; STEPS=${2:-8}
export TMPDIR=/tmp; REPO=$(pwd); cd /tmp; export PYTHONPATH=$REPO
rm -rf $REPO/gpurun_out/kt_$ENV
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/kt_$ENV -- python3 $REPO/bench.py --env $ENV --steps $STEPS --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $REPO; python3 - "$ENV" <<'PY'
import csv, glob, collections, sys
d = collections.defaultdict(list)
for f in glob.glob('gpurun_out/kt_%s/**/*kernel_trace.csv' % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:7]:
    print("%-28s n=%3d avg %.3f ms max %.3f total %.1f" % (k[:28], len(v), sum(v) / len(v), max(v), sum(v)))
PY
