#!/bin/bash
# round 4, run 5: agents50 tail with and without heavy workgroups (per-thread timing of the drivers)
set -o pipefail
mkdir -p gpurun_out/r4e
for hv in 0 128; do
  export MRP_HL_HEAVY_WGS=$hv
  MRP_HL_TIMING=1 MRP_CAP=400000 MRP_REPS=2 timeout -k 10 300 python scripts/quick_bench.py 16384 50 8 512 > gpurun_out/r4e/a50_h${hv}.log 2>&1 || { echo "failed a50 $hv"; exit 1; }
  echo "== agents50 B=16384 heavy=$hv"; grep "rep 1\|kernel tiers\|busy fraction\|heavy workgroups" gpurun_out/r4e/a50_h${hv}.log | tail -4
  grep "last #0\|loop ended\|host ms" gpurun_out/r4e/a50_h${hv}.log | tail -24
done
