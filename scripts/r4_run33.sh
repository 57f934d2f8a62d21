#!/bin/bash
# round 4, run 33: agents100 with per-worker timing (which instances end the step, when were they admitted, what does the host spend?)
set -o pipefail
mkdir -p gpurun_out/r4ak
MRP_HL_TIMING=1 MRP_CAP=3000000 MRP_REPS=2 timeout -k 10 500 python scripts/quick_bench.py 16384 100 16 512 > gpurun_out/r4ak/a100_timing.log 2>&1 || { echo failed; tail -5 gpurun_out/r4ak/a100_timing.log; exit 1; }
grep "^rep" gpurun_out/r4ak/a100_timing.log
