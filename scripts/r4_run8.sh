#!/bin/bash
# round 4, run 8: agents100 — sixteen single-launch workers vs eight / sixteen pairs (is the leg host-bound?)
set -o pipefail
mkdir -p gpurun_out/r4i
run() {  # name threads
  MRP_CAP=3000000 MRP_REPS=2 timeout -k 10 500 python scripts/quick_bench.py 16384 100 $2 512 > gpurun_out/r4i/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4i/$1.log; exit 1; }
  echo "== $1"; grep "rep \|kernel tiers\|busy fraction\|heavy workgroups\|host thread-seconds" gpurun_out/r4i/$1.log | tail -5
}
MRP_HL_HEAVY_WGS=0 run single16 16
run pair8 8
MRP_HL_PAIR_THREADS=16 run pair16 16
MRP_HL_PAIR_THREADS=12 run pair12 12
