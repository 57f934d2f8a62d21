#!/bin/bash
for sp in 2 4 8; do
  echo "== agents100 B=4096 MRP_HL_SPEC=$sp"; MRP_HL_SPEC=$sp MRP_REPS=1 MRP_CAP=3000000 timeout -k 10 300 python scripts/quick_bench.py 4096 100 16 2>&1 | grep "^rep 0\|busy fraction"
done
for sp in 2 4 8; do
  echo "== agents50 B=16384 MRP_HL_SPEC=$sp"; MRP_HL_SPEC=$sp MRP_REPS=2 MRP_CAP=400000 timeout -k 10 300 python scripts/quick_bench.py 16384 50 16 2>&1 | grep "^rep 1\|busy fraction" | tail -2
done
