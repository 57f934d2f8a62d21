#!/bin/bash
# SIPP with device-resident tables: resident workgroups per worker thread, instance counts
for w in 64 96 128 192; do
  echo "== MRP_HL_SIPP_WGS=$w"
  MRP_HL_SIPP_WGS=$w timeout -k 10 200 python scripts/sipp_bench.py 100 2048 16 0 2>&1 | grep "^rep"
done
echo "== 8192 instances wgs 96 / 128"
MRP_HL_SIPP_WGS=96 timeout -k 10 200 python scripts/sipp_bench.py 100 8192 16 0 2>&1 | grep "^rep"
MRP_HL_SIPP_WGS=128 timeout -k 10 200 python scripts/sipp_bench.py 100 8192 16 0 2>&1 | grep "^rep"
