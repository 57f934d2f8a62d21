#!/bin/bash
# round 4, run 2: front + heavy workgroups — GPU tests, then quick benches at B = 262144 / 16384 with and without them
set -o pipefail
mkdir -p gpurun_out/r4b
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4b/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -15 gpurun_out/r4b/pytest.log
[ $rc -eq 0 ] || exit $rc
for hv in default 0; do
  for B in 262144 16384; do
    if [ $hv = 0 ]; then export MRP_HL_HEAVY_WGS=0; else unset MRP_HL_HEAVY_WGS; fi
    MRP_REPS=3 timeout -k 10 300 python scripts/quick_bench.py $B 10 16 192 > gpurun_out/r4b/quick_${B}_heavy_${hv}.log 2>&1 || { echo "quick $B $hv failed"; tail -5 gpurun_out/r4b/quick_${B}_heavy_${hv}.log; exit 1; }
    echo "== B=$B heavy=$hv"; grep "rep \|kernel tiers\|busy fraction\|heavy workgroups" gpurun_out/r4b/quick_${B}_heavy_${hv}.log | tail -8
  done
done
