#!/bin/bash
for h in 0 3; do
  echo "== agents100 B=4096 helpers=$h"; MRP_HL_HELPERS=$h MRP_REPS=1 MRP_CAP=3000000 timeout -k 10 300 python scripts/quick_bench.py 4096 100 16 2>&1 | grep "^rep 0\|busy frac" | cut -c1-200
done
for h in 0 3 6; do
  echo "== agents100 B=16384 helpers=$h"; MRP_HL_HELPERS=$h MRP_REPS=1 MRP_CAP=3000000 timeout -k 10 300 python scripts/quick_bench.py 16384 100 16 2>&1 | grep "^rep 0\|busy frac" | cut -c1-200
done
for h in 0 3; do
  echo "== agents50 B=65536 helpers=$h"; MRP_HL_HELPERS=$h MRP_REPS=1 MRP_CAP=400000 timeout -k 10 300 python scripts/quick_bench.py 65536 50 16 2>&1 | grep "^rep 0\|busy frac" | cut -c1-200
done
for h in 0 2; do
  echo "== agents10 B=131072 helpers=$h"; MRP_HL_HELPERS=$h MRP_REPS=2 timeout -k 10 300 python scripts/quick_bench.py 131072 10 16 2>&1 | grep "^rep 1\|busy frac" | tail -2 | cut -c1-200
done
