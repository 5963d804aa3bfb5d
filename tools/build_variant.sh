#!/bin/bash
# Experimental library build with extra -D flags: tools/build_variant.sh NAME "-DBP_V_X=1 ..."  ->  benchpush_amd/libbenchpush_hip_NAME.so
# (load it with BP_PROF=1 BP_PROF_LIB=<path>, as tools/ab_bench.sh does)
cd "$(dirname "$0")/../benchpush_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fPIC -shared -std=c++17 \
  -Wno-unused-value $2 -Rpass-analysis=kernel-resource-usage -o ../libbenchpush_hip_$1.so bp_capi.hip 2>&1 | grep -A8 "Function Name: _Z14k_physics_step9DevParams" | grep -E "VGPRs:|Spill|Occupancy" | tr '\n' ' '; echo " <- $1"
