#!/bin/bash
# Instruction / phase attribution of the step kernel (run on the GPU box): tools/profile_insts.sh TAG BASE_LIB BASE_PROF_LIB
#   -> gpurun_out/insts_TAG/{phases_new.txt, phases_base.txt, pmc_ab.txt}
# PC sampling and thread trace are not available on this pool (rocprofv3-avail lists no PC-sampling agent; the ATT decoder library is not installed),
# so the attribution is: per-phase wave cycles and trip counts from the BP_PROF build (s_memtime stamps in LDS, tools/prof_phases.py), the kernel's
# instruction mix per launch from the SQ counters (tools/pmc_ab.sh) and the static instruction count of every phase from the assembly
# (tools/isa_blocks.py on a -gline-tables-only build).
TAG=${1:-r03}; BASE=$2; BASEPROF=$3
OUT=gpurun_out/insts_$TAG; mkdir -p $OUT
BP_SCHED=0 BP_PROF=1 python tools/prof_phases.py 4096 26 2>&1 | tail -18 > $OUT/phases_new.txt
if [ -n "$BASEPROF" ]; then BP_SCHED=0 BP_PROF_LIB=$BASEPROF python tools/prof_phases24.py 4096 26 2>&1 | tail -4 > $OUT/phases_base.txt; fi
if [ -n "$BASE" ]; then tools/pmc_ab.sh $BASE $PWD/benchpush_amd/libbenchpush_hip.so > $OUT/pmc_ab.txt 2>&1; else tools/pmc_ab.sh $PWD/benchpush_amd/libbenchpush_hip.so > $OUT/pmc_ab.txt 2>&1; fi
cat $OUT/phases_new.txt $OUT/pmc_ab.txt
