#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the default bench.py + separate PMC passes (kernel-trace only, as the pool requires).
#   tools/profile_gpu.sh TAG [BUILD_ID]   -> gpurun_out/prof_TAG/{summary.txt, pmc.json, kernel_stats.csv, bench_under_rocprof.json}
TAG=${1:-r02}
BUILD=${2:-unknown}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export PYTHONPATH=$REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
PMCARGS="--steps 6 --warmup 24 --no-cpu-baseline --no-steady-state"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $PMCARGS > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $PMCARGS > /dev/null 2> $OUT/pmc_write.log
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $PMCARGS > /dev/null 2> $OUT/pmc_sq.log
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/bench.py $PMCARGS > /dev/null 2> $OUT/pmc_sq2.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN --output-format csv -d $OUT/pmc_lds -- python3 $REPO/bench.py $PMCARGS > /dev/null 2> $OUT/pmc_lds.log
cd $REPO
python3 tools/summarize_prof.py $OUT $BUILD > $OUT/summary.txt 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
cat $OUT/summary.txt
