R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for b in 262144 16384; do
  MRP_REPS=3 timeout -k 5 200 python scripts/quick_bench.py $b 10 16 512 2>&1 | grep "^rep 2\|kernel tiers\|busy fraction" | tail -3
done
timeout -k 10 300 python -m pytest tests/test_ll_parity_gpu.py tests/test_hl_parity_gpu.py -m gpu -x -q 2>&1 | tail -2
