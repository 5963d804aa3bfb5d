#!/bin/bash
# Same-box kernel traces of the default bench.py with the preemptive scheduler (default) and without (BP_SCHED=0): per-launch times of the step kernel
export TMPDIR=/tmp; REPO=$(pwd); cd /tmp; export PYTHONPATH=$REPO
for mode in sched nosched; do
  rm -rf $REPO/gpurun_out/kt_$mode
  if [ $mode = nosched ]; then export BP_SCHED=0; else unset BP_SCHED; fi
  rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/kt_$mode -- python3 $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
done
cd $REPO; python3 - <<'PY'
import csv, glob
for mode in ("sched", "nosched"):
    rows = []
    for f in glob.glob('gpurun_out/kt_%s/**/*kernel_trace.csv' % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name'].split('(')[0]
            if n.startswith('k_physics_step'): rows.append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, n))
    rows.sort()
    du = [d for _, d, _ in rows]
    print("%-8s %s: %d launches; timed region (5..34) avg %.3f ms; steady-state region (last 30) avg %.3f ms" % (mode, rows[0][2], len(du), sum(du[5:35]) / 30, sum(du[-30:]) / 30))
    print("   per launch, steps 0..39: " + " ".join("%.1f" % x for x in du[:40]))
PY
