#!/bin/bash
# round 4, run 11: front capacity beside heavy workgroups of 31 KB / 21 KB windows
set -o pipefail
mkdir -p gpurun_out/r4l
for v in wide12 wide8; do for w in 324 344; do
  LD_PRELOAD=$PWD/libmultirobotplanning_amd/lib/libmrp_ll_$v.so MRP_HL_SESSION_WGS=$w MRP_HL_TIMING=1 MRP_REPS=2 timeout -k 10 300 python scripts/quick_bench.py 262144 10 8 512 > gpurun_out/r4l/${v}_w$w.log 2>&1
  echo "== $v front workgroups per worker $w"
  grep "rep 1\|kernel tiers" gpurun_out/r4l/${v}_w$w.log | cut -c1-150
  grep "group of" gpurun_out/r4l/${v}_w$w.log | tail -8 | sed 's/.*group of \([0-9]*\):.*session_begin \([0-9.]*\) ms.*active wgs \([0-9]*\),.*heavy wgs \([0-9]*\) .*/  instances \1 begin_ms \2 cumulative_active \3 heavy \4/' | tr '\n' ';'; echo
done; done
