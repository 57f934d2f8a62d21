#!/bin/bash
# round 4, run 20: root chains cut into jobs of at most MRP_HL_CHAIN_CHUNK searches — agents50 / agents100 legs
set -o pipefail
mkdir -p gpurun_out/r4x
timeout -k 10 600 python -m pytest tests/test_hl_parity_gpu.py -m gpu -x -q -k "not heavy_tail" > gpurun_out/r4x/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4x/pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert" gpurun_out/r4x/pytest.log | head; exit $rc; }
run() {  # name agents B cap
  MRP_CAP=$4 MRP_REPS=3 timeout -k 10 500 python scripts/quick_bench.py $3 $2 16 512 > gpurun_out/r4x/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4x/$1.log; exit 1; }
  echo "== $1"; grep "rep 2\|busy fraction" gpurun_out/r4x/$1.log | tail -2
}
MRP_HL_CHAIN_CHUNK=16 run a50_c16 50 65536 400000
MRP_HL_CHAIN_CHUNK=8 run a50_c8 50 65536 400000
MRP_HL_CHAIN_CHUNK=4 run a50_c4 50 65536 400000
MRP_HL_CHAIN_CHUNK=8 run a100_c8 100 16384 3000000
MRP_HL_ROOT_CHAIN=0 run a100_nochain 100 16384 3000000
