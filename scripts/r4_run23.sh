#!/bin/bash
# round 4, run 23: SIPP legs, the previous table layout (ab_old/ = the last commit) against the 128-byte records, same box
set -o pipefail
mkdir -p gpurun_out/r4aa
run() {  # name dir agents n
  ( cd $2 && MRP_NO_CPU=1 timeout -k 10 300 python scripts/sipp_bench.py $3 $4 16 0 ) > gpurun_out/r4aa/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4aa/$1.log; exit 1; }
  echo "== $1"; grep "rep 1\|SIPP kernel" gpurun_out/r4aa/$1.log | tail -2
}
run old_s100 ab_old 100 8192
run new_s100 . 100 8192
run old_s200 ab_old 200 4096
run new_s200 . 200 4096
run old_s100b ab_old 100 8192
run new_s100b . 100 8192
