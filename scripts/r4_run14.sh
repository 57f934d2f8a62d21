#!/bin/bash
# round 4, run 14: conflict-free roots written without a conflict tree — GPU tests, then the host-thread sweep
set -o pipefail
mkdir -p gpurun_out/r4p
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4p/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4p/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert" gpurun_out/r4p/pytest.log | head -20; exit $rc; }
run() {  # name B threads slots
  MRP_REPS=3 timeout -k 10 500 python scripts/quick_bench.py $2 10 $3 $4 > gpurun_out/r4p/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4p/$1.log; exit 1; }
  echo "== $1"; grep "rep 2\|busy fraction\|host thread-seconds" gpurun_out/r4p/$1.log | tail -3
}
run w16 262144 16 512
MRP_HL_ROOT_FAST=0 run w16_nofast 262144 16 512
run w8 262144 8 512
run w4 262144 4 1024
MRP_HL_ROOT_FAST=0 run w4_nofast 262144 4 1024
run w2 262144 2 1536
MRP_HL_ROOT_FAST=0 run w2_nofast 262144 2 1536
