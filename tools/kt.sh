export TMPDIR=/tmp; REPO=$(pwd); cd /tmp; export PYTHONPATH=$REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/kt -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $REPO; python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob('gpurun_out/kt/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'].split('(')[0]].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for k, v in sorted(d.items(), key=lambda kv: -sum(b-a for a,b in kv[1]))[:6]:
    du = [(b-a)/1e6 for a,b in v]
    print(k[:40], len(v), 'avg %.3f' % (sum(du)/len(du)), 'last5', ['%.2f' % x for x in du[-5:]])
PY
