#!/bin/bash
# round 4, call 1: solver-pass floor microbenchmark, same-box bench of the round-3 build, LDS counter pass
REPO=$(pwd); OUT=$REPO/gpurun_out/r04_c1; mkdir -p $OUT
timeout -k 10 300 tools/micro/solver_pass > $OUT/solver_pass.txt 2>&1 || exit 1
python bench.py --no-cpu-baseline > $OUT/bench_base.json 2> $OUT/bench_base.err || exit 1
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*" | sort -u > $OUT/lds_counters.txt
PMC_KERNELS=k_physics_step_sched bash tools/pmc_sq.sh 4096 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY > $OUT/pmc_lds.txt 2>&1
cat $OUT/solver_pass.txt $OUT/bench_base.json $OUT/lds_counters.txt $OUT/pmc_lds.txt
