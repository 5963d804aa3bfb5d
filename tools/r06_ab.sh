#!/bin/bash
# Round-6 same-box A/B of library builds with traffic and instruction counters: tools/r06_ab.sh OUTDIR lib1.so lib2.so ...
#   bench (interleaved, twice), then per library one --pmc pass each for FETCH_SIZE, WRITE_SIZE, the SQ instruction mix and the TCC hit / miss counters
#   (launches 24..29 of fresh episodes at 4096 envs, k_physics_step_schedl).
OUT=$1; shift
REPO=$(pwd); mkdir -p $OUT
[ -n "$R06_SKIP_BENCH" ] || bash tools/ab_libs.sh $OUT/ab_bench.txt "" "$@"
export PMC_KERNELS=${PMC_KERNELS:-k_physics_step_schedl}
for lib in "$@"; do
  lib=$(realpath $lib)   # pmc_sq.sh runs the bench from /tmp
  echo "== $(basename $lib)" >> $OUT/pmc.txt
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU"; do
    BP_PROF=1 BP_PROF_LIB=$lib bash tools/pmc_sq.sh ${R06_ENVS:-4096} $grp >> $OUT/pmc.txt 2>&1
  done
done
echo done >> $OUT/pmc.txt
