#!/bin/bash
echo "== agents100 B=16384"; MRP_REPS=1 MRP_CAP=3000000 timeout -k 10 500 python scripts/quick_bench.py 16384 100 16 2>&1 | grep "^rep 0\|busy fraction\|peak host"
grep -i "vmhwm\|vmpeak" /proc/self/status
echo "== agents50 B=65536"; MRP_REPS=1 MRP_CAP=400000 timeout -k 10 500 python scripts/quick_bench.py 65536 50 16 2>&1 | grep "^rep 0\|busy fraction\|peak host"
