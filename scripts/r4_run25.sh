#!/bin/bash
# round 4, run 25: SIPP legs at 50 and 200 agents, last commit (ab_old/) against the two-row layout
set -o pipefail
mkdir -p gpurun_out/r4ac
run() {  # name dir agents n
  ( cd $2 && MRP_NO_CPU=1 timeout -k 10 300 python scripts/sipp_bench.py $3 $4 16 0 ) > gpurun_out/r4ac/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4ac/$1.log; exit 1; }
  echo "== $1"; grep "rep 1\|SIPP kernel" gpurun_out/r4ac/$1.log | tail -2
}
run old_s200 ab_old 200 4096
run new_s200 . 200 4096
MRP_LL_SIPP_TABLES_UNCACHED=1 run new_s200_unc . 200 4096
run old_s50 ab_old 50 8192
run new_s50 . 50 8192
