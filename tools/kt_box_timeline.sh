#!/bin/bash
# Kernel timeline of the box-delivery bench (run on the GPU box via gpurun): start / end of every k_bd_* launch of the last steps relative to the step's plan kernel,
# to see what runs beside what in the two-pass step.   BP_BD_BUDGET=3000 tools/kt_box_timeline.sh
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp; export PYTHONPATH=$REPO
rm -rf $REPO/gpurun_out/kt_box_tl
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/kt_box_tl -- python3 $REPO/bench.py --env box --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $REPO; python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/kt_box_tl/**/*kernel_trace.csv', recursive=True):
    rows += [r for r in csv.DictReader(open(f))]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0], r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows if r['Kernel_Name'].startswith(('k_bd', 'k_ac'))]
rows.sort()
plans = [i for i, r in enumerate(rows) if r[2] == 'k_bd_plan']
print('step durations (ms, plan start to the last kernel end):', ' '.join('%.1f' % ((max(r[1] for r in rows[a:b]) - rows[a][0]) / 1e6) for a, b in zip(plans, plans[1:] + [len(rows)])))
for a, b in list(zip(plans, plans[1:] + [len(rows)]))[-5:]:
    t0 = rows[a][0]
    print('--- step')
    for s, e, n, q in rows[a:b]:
        print('  %-22s queue %-4s %9.3f .. %9.3f ms  (%.3f)' % (n, q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
PY
