set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for cfg in "" "MRP_HL_PRIO_EXPANSIONS=1" "MRP_HL_ACTIVE_LIMIT=1536" "MRP_HL_RING_DEPTH=160" "MRP_HL_RING_DEPTH=336"; do
  for B in 16384 65536; do
    echo "== [$cfg] B=$B"
    env $cfg MRP_REPS=3 timeout -k 5 200 python scripts/quick_bench.py $B 10 16 2>&1 | grep "^rep\|busy fraction" | tail -4
  done
done
