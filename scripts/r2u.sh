#!/bin/bash
# compact-tier geometry sweep with 10-bit node ids (LD_PRELOAD of the A/B build)
V=$PWD/libmultirobotplanning_amd/lib/libmrp_ll_id10.so
run() { echo "== $1 tier=$2"; LD_PRELOAD=$3 MRP_HL_TIER=$2 MRP_REPS=2 timeout -k 10 300 python scripts/quick_bench.py 131072 10 16 2>&1 | grep -A5 "^rep 1" | grep "^rep\|tiers\|busy frac"; }
run base 400,48,2048 ""
run id10 400,48,2048 $V
run id10 512,48,2048 $V
run id10 640,56,2048 $V
run id10 768,64,2048 $V
run id10 1024,64,2048 $V
