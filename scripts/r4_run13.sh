#!/bin/bash
# round 4, run 13: co-workers — sixteen worker threads on eight engines
set -o pipefail
mkdir -p gpurun_out/r4n
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4n/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4n/pytest.log
[ $rc -eq 0 ] || exit $rc
run() {  # name agents B cap threads
  MRP_CAP=$4 MRP_REPS=3 timeout -k 10 500 python scripts/quick_bench.py $3 $2 $5 512 > gpurun_out/r4n/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4n/$1.log; exit 1; }
  echo "== $1"; grep "rep 2\|kernel tiers\|busy fraction\|heavy workgroups\|host thread-seconds" gpurun_out/r4n/$1.log | tail -5
}
run a10_w16 10 262144 50000 16
run a10_w8 10 262144 50000 8
run a10_B16384_w16 10 16384 50000 16
run a50_w16 50 65536 400000 16
run a50_w8 50 65536 400000 8
run a100_w16 100 16384 3000000 16
