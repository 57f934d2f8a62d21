R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for a in 0 16384 8192; do
  echo "== MRP_ARENA_NODES=$a"
  MRP_ARENA_NODES=$a MRP_REPS=3 timeout -k 5 200 python scripts/quick_bench.py 262144 10 16 512 2>&1 | grep "^rep 2\|kernel tiers\|busy fraction" | tail -3
done
