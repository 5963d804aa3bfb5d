#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for HBM traffic. Outputs -> gpurun_out/prof_<tag>/
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export PYTHONPATH=$REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.log
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq.log
cd $REPO
find $OUT -name "*.csv" | head -40
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
