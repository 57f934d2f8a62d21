#!/bin/bash
# round 4, run 21: SIPP resident tables as one 128-byte record per cell — parity tests, then the three legs
set -o pipefail
mkdir -p gpurun_out/r4y
timeout -k 10 900 python -m pytest tests/test_ll_parity_gpu.py tests/test_hl_parity_gpu.py -m gpu -x -q -k "sipp" > gpurun_out/r4y/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4y/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert" gpurun_out/r4y/pytest.log | head -20; exit $rc; }
run() {  # name agents n
  MRP_NO_CPU=1 timeout -k 10 300 python scripts/sipp_bench.py $2 $3 16 0 > gpurun_out/r4y/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4y/$1.log; exit 1; }
  echo "== $1"; grep "rep 1\|SIPP kernel" gpurun_out/r4y/$1.log | tail -2
}
run s100 100 8192
MRP_LL_SIPP_TABLES_UNCACHED=1 run s100_uncached 100 8192
run s50 50 8192
run s200 200 4096
