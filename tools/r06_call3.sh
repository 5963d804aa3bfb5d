#!/bin/bash
# round 6, GPU call 3: counters of base vs no-wn, pairing / batch-size A/B of the two, three waves per SIMD, the new fuzz test
OUT=gpurun_out/r06_c3; mkdir -p $OUT
python -m pytest tests/test_gpu_fuzz.py -x -q > $OUT/fuzz.log 2>&1; tail -3 $OUT/fuzz.log
B=benchpush_amd/libbenchpush_hip_base.so; N=benchpush_amd/libbenchpush_hip_nown.so; W=benchpush_amd/libbenchpush_hip_w3.so
R06_SKIP_BENCH=1 bash tools/r06_ab.sh $OUT $B $N
# pairing at 4096 envs and the throughput regime, base vs no-wn (same box, interleaved twice)
BP_PAIR=2 bash tools/ab_libs.sh $OUT/ab_pair4096.txt "--no-steady-state" $B $N
bash tools/ab_libs.sh $OUT/ab_8192.txt "--envs-per-gpu 8192 --no-steady-state" $B $N
bash tools/ab_libs.sh $OUT/ab_16384.txt "--envs-per-gpu 16384 --no-steady-state --steps 20" $B $N
# three waves per SIMD on the no-wn code (168 VGPRs, 40 velocity slots, 64 queries); capacities are shrunk on purpose
BP_BENCH_IGNORE_CAPACITY=1 bash tools/ab_libs.sh $OUT/ab_w3.txt "--no-steady-state" $N $W
BP_BENCH_IGNORE_CAPACITY=1 bash tools/ab_libs.sh $OUT/ab_w3_12288.txt "--envs-per-gpu 12288 --no-steady-state --steps 20" $N $W
echo done > $OUT/done
