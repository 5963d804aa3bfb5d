#!/bin/bash
# PC sampling of the default bench.py on the line-table build (benchpush_amd/libbenchpush_hip_g.so = build.py's flags + -gline-tables-only):
#   tools/pcsample.sh TAG [method] [interval]   ->  gpurun_out/pcs_TAG/   (summarised by tools/pcsample_report.py)
TAG=${1:-r03}; METHOD=${2:-stochastic}; INTERVAL=${3:-1048576}
REPO=$(pwd); OUT=$REPO/gpurun_out/pcs_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp; export PYTHONPATH=$REPO
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
UNIT=cycles; if [ "$METHOD" = "host_trap" ]; then UNIT=time; fi
BP_PROF=1 BP_PROF_LIB=$REPO/benchpush_amd/libbenchpush_hip_g.so timeout -k 10 400 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT \
  --pc-sampling-interval $INTERVAL --kernel-trace --output-format csv -d $OUT/raw -- python3 $REPO/bench.py --steps 6 --warmup 24 --no-cpu-baseline --no-steady-state \
  > $OUT/bench.json 2> $OUT/log.txt
rc=$?
echo "pcsample rc=$rc"; tail -5 $OUT/log.txt; find $OUT/raw -type f | head; du -sh $OUT
exit $rc
