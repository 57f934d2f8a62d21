#!/bin/bash
# round 4, run 4: how many heavy workgroups?  8 worker threads (16 resident kernels)
set -o pipefail
mkdir -p gpurun_out/r4d
for hv in 96 128 192 256; do
  export MRP_HL_HEAVY_WGS=$hv
  MRP_REPS=3 timeout -k 10 300 python scripts/quick_bench.py 262144 10 8 512 > gpurun_out/r4d/q262144_h${hv}.log 2>&1 || { echo "failed $hv"; tail -5 gpurun_out/r4d/q262144_h${hv}.log; exit 1; }
  echo "== B=262144 heavy=$hv"; grep "rep \|kernel tiers\|busy fraction\|heavy workgroups" gpurun_out/r4d/q262144_h${hv}.log | tail -4
done
for hv in 0 128 256; do
  export MRP_HL_HEAVY_WGS=$hv
  MRP_REPS=3 timeout -k 10 300 python scripts/quick_bench.py 16384 10 8 512 > gpurun_out/r4d/q16384_h${hv}.log 2>&1 || { echo "failed $hv"; exit 1; }
  echo "== B=16384 heavy=$hv"; grep "rep \|kernel tiers\|busy fraction\|heavy workgroups" gpurun_out/r4d/q16384_h${hv}.log | tail -4
  MRP_CAP=400000 MRP_REPS=2 timeout -k 10 300 python scripts/quick_bench.py 16384 50 8 512 > gpurun_out/r4d/a50_h${hv}.log 2>&1 || { echo "failed a50 $hv"; exit 1; }
  echo "== agents50 B=16384 heavy=$hv"; grep "rep \|kernel tiers\|busy fraction\|heavy workgroups" gpurun_out/r4d/a50_h${hv}.log | tail -4
done
