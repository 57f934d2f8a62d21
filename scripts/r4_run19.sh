#!/bin/bash
# round 4, run 19: root chains up to 128 agents — HL parity tests, agents50 / agents100 legs
set -o pipefail
mkdir -p gpurun_out/r4w
timeout -k 10 900 python -m pytest tests/test_hl_parity_gpu.py tests/test_ll_parity_gpu.py -m gpu -x -q -k "not heavy_tail" > gpurun_out/r4w/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4w/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert" gpurun_out/r4w/pytest.log | head; exit $rc; }
run() {  # name agents B cap
  MRP_CAP=$4 MRP_REPS=3 timeout -k 10 500 python scripts/quick_bench.py $3 $2 16 512 > gpurun_out/r4w/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4w/$1.log; exit 1; }
  echo "== $1"; grep "rep 2\|kernel tiers\|busy fraction\|host thread-seconds" gpurun_out/r4w/$1.log | tail -4
}
run a50 50 65536 400000
MRP_HL_ROOT_CHAIN=0 run a50_nochain 50 65536 400000
run a100 100 16384 3000000
