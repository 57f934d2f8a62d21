#!/bin/bash
V=$PWD/libmultirobotplanning_amd/lib/libmrp_ll_fence.so
run() { echo "== $1"; LD_PRELOAD=$3 MRP_HL_STORE_MAX_AGENTS=$2 MRP_REPS=1 MRP_CAP=3000000 timeout -k 10 300 python scripts/quick_bench.py 4096 100 16 2>&1 | grep "^rep 0\|tiers\|staged" | cut -c1-200; }
run "agents100 tables (no store)" 64 ""
run "agents100 store, agent-scope loads" 128 ""
run "agents100 store, acquire fence + plain loads" 128 $V
