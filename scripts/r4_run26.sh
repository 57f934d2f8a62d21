#!/bin/bash
# round 4, run 26: the headline shape with per-worker timing (where does the step end?)
set -o pipefail
mkdir -p gpurun_out/r4ad
MRP_HL_TIMING=1 MRP_CAP=50000 MRP_REPS=3 timeout -k 10 500 python scripts/quick_bench.py 262144 10 16 512 > gpurun_out/r4ad/timing.log 2>&1 || { echo failed; tail -5 gpurun_out/r4ad/timing.log; exit 1; }
grep -v "^\[chain\]" gpurun_out/r4ad/timing.log | grep "rep 2\|busy fraction\|heavy workgroups" | tail -3
