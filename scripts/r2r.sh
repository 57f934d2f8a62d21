#!/bin/bash
echo "== agents10 B=131072"; MRP_REPS=2 timeout -k 10 300 python scripts/quick_bench.py 131072 10 16 2>&1 | grep -A7 "^rep 1"
echo "== agents50 B=16384"; MRP_REPS=2 MRP_CAP=400000 timeout -k 10 300 python scripts/quick_bench.py 16384 50 16 2>&1 | grep -A7 "^rep 1"
echo "== agents100 B=4096"; MRP_REPS=1 MRP_CAP=3000000 timeout -k 10 300 python scripts/quick_bench.py 4096 100 16 2>&1 | grep -A7 "^rep 0"
