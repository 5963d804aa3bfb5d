#!/bin/bash
# kernel-trace of the box-delivery bench with the searches of k_bd_finish / k_bd_robot_map in LDS (default where they fit) and through the L2 (BP_BD_LDIST=0)
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp; export PYTHONPATH=$REPO
for v in 1 0; do
  rm -rf $REPO/gpurun_out/kt_box_ld$v
  BP_BD_LDIST=$v rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/kt_box_ld$v -- python3 $REPO/bench.py --env ${KT_ENV:-box} --steps 8 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/kt_box_ld$v.log 2>&1
  python3 - $REPO/gpurun_out/kt_box_ld$v $v <<'PY'
import csv,glob,collections,sys
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
print("BP_BD_LDIST=%s" % sys.argv[2])
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    if sum(v)>1: print('  %-28s n=%d avg=%.2f min=%.2f max=%.2f' % (k[:28],len(v),sum(v)/len(v),min(v),max(v)))
PY
done
