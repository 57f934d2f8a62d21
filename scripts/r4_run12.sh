#!/bin/bash
# round 4, run 12: heavy workgroup count with the 31 KB wide window (eight workers); GPU tests first
set -o pipefail
mkdir -p gpurun_out/r4m
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4m/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4m/pytest.log
[ $rc -eq 0 ] || exit $rc
run() {  # name agents B cap
  MRP_HL_TIMING=1 MRP_CAP=$4 MRP_REPS=3 timeout -k 10 500 python scripts/quick_bench.py $3 $2 8 512 > gpurun_out/r4m/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4m/$1.log; exit 1; }
  echo "== $1"; grep "rep 2\|kernel tiers\|busy fraction\|heavy workgroups" gpurun_out/r4m/$1.log | tail -4
  grep "group of" gpurun_out/r4m/$1.log | tail -8 | sed 's/.*active wgs \([0-9]*\),.*/\1/' | tr '\n' ' '; echo
}
for hv in 160 192 224 256; do export MRP_HL_HEAVY_WGS=$hv; run a10_h${hv} 10 262144 50000; done
for hv in 192 256; do export MRP_HL_HEAVY_WGS=$hv; run a50_h${hv} 50 65536 400000; done
unset MRP_HL_HEAVY_WGS; run a10_B16384 10 16384 50000
