#!/bin/bash
# smaller compact tier, more resident wavefronts (hybrid arena tier makes migrations cheaper)
run() { echo "== tier=$1 wgs/thread=$2"; MRP_HL_TIER=$1 MRP_REPS=2 timeout -k 10 300 python scripts/quick_bench.py 131072 10 16 2>&1 | grep -A5 "^rep 1" | grep "^rep\|tiers\|busy frac"; }
run 400,48,2048 0
run 352,44,2048 0
run 320,40,1536 0
run 256,36,1536 0
run 192,32,1024 0
