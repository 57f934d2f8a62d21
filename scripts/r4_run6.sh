#!/bin/bash
# round 4, run 6: wide tier with long horizons; worker-thread cap; heavy workgroup count
set -o pipefail
mkdir -p gpurun_out/r4f
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4f/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -5 gpurun_out/r4f/pytest.log
[ $rc -eq 0 ] || exit $rc
run() {  # name agents B cap
  MRP_CAP=$4 MRP_REPS=3 timeout -k 10 400 python scripts/quick_bench.py $3 $2 16 512 > gpurun_out/r4f/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4f/$1.log; exit 1; }
  echo "== $1"; grep "rep \|kernel tiers\|busy fraction\|heavy workgroups" gpurun_out/r4f/$1.log | tail -4
}
for pt in 8 10; do for hv in 160 192 224; do
  export MRP_HL_PAIR_THREADS=$pt MRP_HL_HEAVY_WGS=$hv
  run a10_t${pt}_h${hv} 10 262144 50000
done; done
export MRP_HL_PAIR_THREADS=8
for hv in 0 128 256; do
  export MRP_HL_HEAVY_WGS=$hv
  run a50_h${hv} 50 16384 400000
done
