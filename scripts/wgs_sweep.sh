# Resident workgroups per engine against throughput and per-expansion times (GPU box): 16 engines x 112 / 144 / 192
# workgroups = 7 / 9 / 12 searches per CU (the A*-epsilon kernels' LDS window allows 12).
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for w in 112 144 192; do
  echo "== MRP_HL_SESSION_WGS=$w"
  MRP_HL_SESSION_WGS=$w MRP_REPS=3 timeout -k 5 200 python scripts/quick_bench.py 262144 10 16 512 2>&1 | grep "^rep 2\|kernel tiers\|busy fraction\|host thread" | tail -4
done
