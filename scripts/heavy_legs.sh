#!/bin/bash
# the host-heavy ECBS shapes (dev probe): agents100 / agents50 through quick_bench with the host time breakdown
echo "== agents100 B=4096"; MRP_HL_TIMING=1 MRP_REPS=1 MRP_CAP=3000000 timeout -k 10 300 python scripts/quick_bench.py 4096 100 16 2>&1 | grep "^rep 0\|busy frac\|host ms" | head -4 | cut -c1-220
echo "== agents100 B=16384"; MRP_REPS=1 MRP_CAP=3000000 timeout -k 10 300 python scripts/quick_bench.py 16384 100 16 2>&1 | grep "^rep 0\|busy frac" | cut -c1-200
echo "== agents50 B=65536"; MRP_REPS=1 MRP_CAP=400000 timeout -k 10 300 python scripts/quick_bench.py 65536 50 16 2>&1 | grep "^rep 0\|busy frac" | cut -c1-200
