#!/bin/bash
# round 4, run 24: SIPP resident tables as a 64-byte bounds row + a 64-byte status row per cell, against the last commit (ab_old/)
set -o pipefail
mkdir -p gpurun_out/r4ab
timeout -k 10 900 python -m pytest tests/test_ll_parity_gpu.py tests/test_hl_parity_gpu.py -m gpu -x -q -k "sipp" > gpurun_out/r4ab/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4ab/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert" gpurun_out/r4ab/pytest.log | head -20; exit $rc; }
run() {  # name dir agents n
  ( cd $2 && MRP_NO_CPU=1 timeout -k 10 300 python scripts/sipp_bench.py $3 $4 16 0 ) > gpurun_out/r4ab/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4ab/$1.log; exit 1; }
  echo "== $1"; grep "rep 1\|SIPP kernel" gpurun_out/r4ab/$1.log | tail -2
}
run old_s100 ab_old 100 8192
run new_s100 . 100 8192




MRP_LL_SIPP_TABLES_UNCACHED=1 run new_s100_unc . 100 8192
