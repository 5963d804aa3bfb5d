export TMPDIR=/tmp; REPO=$(pwd); cd /tmp; export PYTHONPATH=$REPO
rm -rf $REPO/gpurun_out/kt_mix
BP_MIX=512 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/kt_mix -- python3 $REPO/bench.py --steps 6 --warmup 20 --no-cpu-baseline --no-steady-state > /dev/null 2>&1
cd $REPO; python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob('gpurun_out/kt_mix/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0]
        if n.startswith('k_physics') or n.startswith('k_delay'):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n))
rows.sort()
t0=rows[0][0]
for s,e,n in rows[-12:]:
    print("%-26s start %10.3f ms  end %10.3f ms  dur %7.3f" % (n, (s-t0)/1e6, (e-t0)/1e6, (e-s)/1e6))
PY
