#!/bin/bash
# round 4, run 22: SIPP legs at 8 worker threads (16 sessions = 32 streams put the device into time-slicing, run 7)
set -o pipefail
mkdir -p gpurun_out/r4z
run() {  # name agents n threads
  MRP_NO_CPU=1 timeout -k 10 300 python scripts/sipp_bench.py $2 $3 $4 0 > gpurun_out/r4z/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4z/$1.log; exit 1; }
  echo "== $1"; grep "rep 1\|SIPP kernel" gpurun_out/r4z/$1.log | tail -2
}
run s100_t8 100 8192 8
run s100_t10 100 8192 10
run s100_t12 100 8192 12
MRP_LL_SIPP_TABLES_UNCACHED=1 run s100_t8_unc 100 8192 8
run s50_t8 50 8192 8
run s200_t8 200 4096 8
