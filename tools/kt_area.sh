#!/bin/bash
# kernel-trace stats of the box-delivery bench (run on the GPU box via gpurun)
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp; export PYTHONPATH=$REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/kt_area -- python3 $REPO/bench.py --env area --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $REPO; python3 tools/summarize_prof.py gpurun_out/kt_area 2>/dev/null | head -14
