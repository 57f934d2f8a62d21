#!/bin/bash
# round 4, run 7: eight engines (16 resident kernels): heavy workgroup count per agent count
set -o pipefail
mkdir -p gpurun_out/r4g
run() {  # name agents B cap
  MRP_CAP=$4 MRP_REPS=3 timeout -k 10 500 python scripts/quick_bench.py $3 $2 8 512 > gpurun_out/r4g/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4g/$1.log; exit 1; }
  echo "== $1"; grep "rep \|kernel tiers\|busy fraction\|heavy workgroups" gpurun_out/r4g/$1.log | tail -4
}
for hv in 160 192; do export MRP_HL_HEAVY_WGS=$hv; run a10_h${hv} 10 262144 50000; done
for hv in 0 192 256; do export MRP_HL_HEAVY_WGS=$hv; run a50_h${hv} 50 16384 400000; done
for hv in 0 256; do export MRP_HL_HEAVY_WGS=$hv; run a50big_h${hv} 50 65536 400000; done
for hv in 0 256; do export MRP_HL_HEAVY_WGS=$hv; run a100_h${hv} 100 4096 3000000; done
