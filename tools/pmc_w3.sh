#!/bin/bash
# counters of the step kernel at two (product) and three (libbenchpush_hip_w3.so, tools/build_variant.sh) waves per SIMD: where does the third wave's time go?
export BP_BENCH_IGNORE_CAPACITY=1 PMC_KERNELS=k_physics_step_sched
for lib in benchpush_amd/libbenchpush_hip.so benchpush_amd/libbenchpush_hip_w3.so; do
  echo "== $lib"
  export BP_PROF=1 BP_PROF_LIB=$(pwd)/$lib
  bash tools/pmc_sq.sh 4096 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
  bash tools/pmc_sq.sh 4096 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
  bash tools/pmc_sq.sh 4096 TCC_HIT_sum TCC_MISS_sum
  bash tools/pmc_sq.sh 4096 SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM
done
