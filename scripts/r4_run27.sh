#!/bin/bash
# round 4, run 27: depth of the device queue at the headline shape (jobs are root chains of ~1.4 ms now, not single searches)
set -o pipefail
mkdir -p gpurun_out/r4ae
for d in 0 260 220 190 175 165; do
  MRP_HL_RING_DEPTH=$d MRP_CAP=50000 MRP_REPS=4 timeout -k 10 300 python scripts/quick_bench.py 262144 10 16 512 > gpurun_out/r4ae/d$d.log 2>&1 || { echo failed $d; tail -5 gpurun_out/r4ae/d$d.log; exit 1; }
  echo "== depth $d: $(grep '^rep' gpurun_out/r4ae/d$d.log | awk '{print $4}' | tr '\n' ' ') $(grep 'busy fraction' gpurun_out/r4ae/d$d.log | tail -1 | awk '{print $NF}')"
done
