#!/bin/bash
# round 4, run 10: how many front workgroups really fit beside 160 heavy ones (eight workers)?
set -o pipefail
mkdir -p gpurun_out/r4k
for w in 270 285 300 312; do
  MRP_HL_SESSION_WGS=$w MRP_HL_TIMING=1 MRP_REPS=2 timeout -k 10 300 python scripts/quick_bench.py 262144 10 8 512 > gpurun_out/r4k/w$w.log 2>&1
  echo "== front workgroups per worker $w"
  grep "rep 1" gpurun_out/r4k/w$w.log | cut -c1-90
  grep "group of" gpurun_out/r4k/w$w.log | tail -8 | sed 's/.*group of \([0-9]*\):.*session_begin \([0-9.]*\) ms.*active wgs \([0-9]*\),.*heavy wgs \([0-9]*\) .*/  instances \1 begin_ms \2 cumulative_active \3 heavy \4/'
done
