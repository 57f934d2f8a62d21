#!/bin/bash
# round 4, run 35: SIPP with uncached tables and no fences — does residency pay now?  LDS tier of 2048 / 1536 / 1024 nodes
# = 6 / 8 / 12 searches per CU (variant libraries built with -DMRP_LL_SIPP_LDS_NODES, preloaded under the drivers)
set -o pipefail
mkdir -p gpurun_out/r4ao
L=$PWD/libmultirobotplanning_amd/lib
run() {  # name preload wgs agents n
  ( [ -n "$2" ] && export LD_PRELOAD=$2; [ -n "$3" ] && export MRP_HL_SIPP_WGS=$3; MRP_NO_CPU=1 timeout -k 10 300 python scripts/sipp_bench.py $4 $5 16 0 ) > gpurun_out/r4ao/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4ao/$1.log; exit 1; }
  echo "== $1: $(grep 'rep 1' gpurun_out/r4ao/$1.log | cut -c1-95) | $(grep 'SIPP kernel' gpurun_out/r4ao/$1.log | tail -1 | cut -c17-50) | $(grep 'middle tier' gpurun_out/r4ao/$1.log | tail -1 | cut -c58-120)"
}
for a in 100 200; do
  n=$([ $a = 200 ] && echo 4096 || echo 8192)
  run s${a}_2048 "" "" $a $n
  run s${a}_1536 $L/libmrp_ll_sipp1536.so 256 $a $n
  run s${a}_1024 $L/libmrp_ll_sipp1024.so 384 $a $n
done
