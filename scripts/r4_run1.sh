#!/bin/bash
# round 4, run 1: GPU tests, quick bench (B=262144, 16 threads, 2 reps), phase profile of the compact tier
set -o pipefail
mkdir -p gpurun_out/r4a
python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4a/pytest.log
tail -3 gpurun_out/r4a/pytest.log
MRP_REPS=3 timeout -k 10 300 python scripts/quick_bench.py 262144 10 16 192 > gpurun_out/r4a/quick_262144.log 2>&1; echo "quick rc=$?"
grep "rep \|kernel tiers\|busy fraction" gpurun_out/r4a/quick_262144.log
MRP_REPS=3 timeout -k 10 200 python scripts/quick_bench.py 16384 10 16 192 > gpurun_out/r4a/quick_16384.log 2>&1; echo "quick16k rc=$?"
grep "rep \|kernel tiers\|busy fraction" gpurun_out/r4a/quick_16384.log
MRP_LL_LIB=libmultirobotplanning_amd/lib/libmrp_ll_ctprof.so timeout -k 10 300 python scripts/ct_phase_profile.py 10 64 1 > gpurun_out/r4a/ct_phase.log 2>&1; echo "phase rc=$?"
cat gpurun_out/r4a/ct_phase.log
