#!/bin/bash
# round 4, run 3: is the front/heavy pair slow because 32 resident kernels oversubscribe the hardware queues?
set -o pipefail
mkdir -p gpurun_out/r4c
for th in 8 12; do
  for hv in default 0; do
    if [ $hv = 0 ]; then export MRP_HL_HEAVY_WGS=0; else unset MRP_HL_HEAVY_WGS; fi
    MRP_REPS=3 timeout -k 10 300 python scripts/quick_bench.py 262144 10 $th 512 > gpurun_out/r4c/quick_t${th}_heavy_${hv}.log 2>&1 || { echo "quick $th $hv failed"; tail -5 gpurun_out/r4c/quick_t${th}_heavy_${hv}.log; exit 1; }
    echo "== threads=$th heavy=$hv"; grep "rep \|kernel tiers\|busy fraction\|heavy workgroups" gpurun_out/r4c/quick_t${th}_heavy_${hv}.log | tail -4
  done
done
