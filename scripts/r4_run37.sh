#!/bin/bash
# round 4, run 37: fifteen searches per CU in the front tier — narrow geometry of 3 groups (767 open entries, 10.1 KB) and a
# front kernel held to 128 VGPRs (four waves per SIMD); variant libraries preloaded under the drivers
set -o pipefail
mkdir -p gpurun_out/r4ar
L=$PWD/libmultirobotplanning_amd/lib
LD_PRELOAD=$L/libmrp_ll_slim.so timeout -k 10 600 python -m pytest tests/test_hl_parity_gpu.py -m gpu -x -q -k "not heavy_tail" > gpurun_out/r4ar/pytest_slim.log 2>&1; rc=$?; echo "pytest (slim) rc=$rc $(tail -1 gpurun_out/r4ar/pytest_slim.log)"
[ $rc -eq 0 ] || { grep -n "Error\|assert" gpurun_out/r4ar/pytest_slim.log | head; exit $rc; }
run() {  # name preload agents B cap
  ( [ -n "$2" ] && export LD_PRELOAD=$2; MRP_CAP=$5 MRP_REPS=4 timeout -k 10 500 python scripts/quick_bench.py $4 $3 16 512 ) > gpurun_out/r4ar/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4ar/$1.log; exit 1; }
  echo "== $1: $(grep '^rep' gpurun_out/r4ar/$1.log | awk '{print $4}' | tr '\n' ' ') | $(grep 'kernel tiers' gpurun_out/r4ar/$1.log | tail -1 | cut -c18-110) | busy $(grep 'busy fraction' gpurun_out/r4ar/$1.log | tail -1 | awk '{print $NF}')"
}
run a10_default "" 10 262144 50000
run a10_waves4 $L/libmrp_ll_waves4.so 10 262144 50000
run a10_slim $L/libmrp_ll_slim.so 10 262144 50000
run a50_default "" 50 65536 400000
run a50_slim $L/libmrp_ll_slim.so 50 65536 400000
