#!/bin/bash
# Same-box counter A/B of library builds (run on the GPU box): tools/pmc_ab.sh lib1.so lib2.so ...  -> SQ instruction mix and wave-time split of the step kernel
# per launch (mean of the last four profiled launches, 4096 envs, steps 26..29 of fresh episodes).  Two rocprofv3 passes per library (8 SQ counters each).
export PMC_KERNELS=${PMC_KERNELS:-k_physics_step_sched}
for lib in "$@"; do
  echo "== $(basename $lib)"
  BP_PROF=1 BP_PROF_LIB=$lib tools/pmc_sq.sh 4096 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
  BP_PROF=1 BP_PROF_LIB=$lib tools/pmc_sq.sh 4096 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
done
