#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4q
MRP_HL_TIMING=1 MRP_REPS=2 timeout -k 10 300 python scripts/quick_bench.py 262144 10 2 1536 > gpurun_out/r4q/w2_timing.log 2>&1
grep "rep 1\|host thread-seconds" gpurun_out/r4q/w2_timing.log | tail -2
grep "host ms\|loop ended\|group of" gpurun_out/r4q/w2_timing.log | tail -6
