#!/bin/bash
for w in 96 144 192; do
  echo "== MRP_HL_SIPP_WGS=$w (512 LDS nodes)"
  MRP_HL_SIPP_WGS=$w timeout -k 10 200 python scripts/sipp_bench.py 100 2048 16 0 2>&1 | grep "^rep\|engine"
done
