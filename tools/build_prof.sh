#!/bin/bash
# Diagnostic build with in-kernel phase timers (loaded by BP_PROF=1): same flags as benchpush_amd/build.py plus -DBP_PROF.
cd "$(dirname "$0")/../benchpush_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fPIC -shared -std=c++17 \
  -Wno-unused-value -DBP_PROF=1 -o ../libbenchpush_hip_prof.so bp_capi.hip
