#!/bin/bash
# round 4, run 9: where the worker threads' time goes at eight workers (agents10 headline shape)
set -o pipefail
mkdir -p gpurun_out/r4j
MRP_HL_TIMING=1 MRP_REPS=2 timeout -k 10 300 python scripts/quick_bench.py 262144 10 8 512 > gpurun_out/r4j/a10_timing.log 2>&1
grep "rep 1\|busy fraction\|host thread-seconds" gpurun_out/r4j/a10_timing.log | tail -3
grep "host ms\|loop ended\|group of" gpurun_out/r4j/a10_timing.log | tail -24
