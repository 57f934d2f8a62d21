#!/bin/bash
# round 4, run 36: SIPP residency, continued: LDS tier of 1024 / 768 / 512 nodes at 12 / 16 / 16 searches per CU
set -o pipefail
mkdir -p gpurun_out/r4ap
L=$PWD/libmultirobotplanning_amd/lib
run() {  # name preload wgs agents n
  ( [ -n "$2" ] && export LD_PRELOAD=$2; [ -n "$3" ] && export MRP_HL_SIPP_WGS=$3; MRP_NO_CPU=1 timeout -k 10 300 python scripts/sipp_bench.py $4 $5 16 0 ) > gpurun_out/r4ap/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4ap/$1.log; exit 1; }
  echo "== $1: $(grep 'rep 1' gpurun_out/r4ap/$1.log | cut -c1-95) | $(grep 'SIPP kernel' gpurun_out/r4ap/$1.log | tail -1 | cut -c17-50) | $(grep 'middle tier' gpurun_out/r4ap/$1.log | tail -1 | cut -c58-120)"
}
for a in 50 100 200; do
  n=$([ $a = 200 ] && echo 4096 || echo 8192)
  run s${a}_1024 $L/libmrp_ll_sipp1024.so 384 $a $n
  run s${a}_768 $L/libmrp_ll_sipp768.so 512 $a $n
  run s${a}_512 $L/libmrp_ll_sipp512.so 512 $a $n
done
