# agents50 / agents100: focal path table in LDS (default: up to 16 KB per search) vs in the arena (2 KB LDS budget)
export MRP_REPS=2
for cfg in "8192 50 400000 none" "8192 50 400000 400,48,2048" "2048 100 2000000 none" "2048 100 2000000 400,48,2048" "65536 10 50000 none"; do
  set -- $cfg
  if [ "$4" = "none" ]; then unset MRP_HL_TIER; else export MRP_HL_TIER=$4; fi
  MRP_CAP=$3 timeout -k 10 250 python scripts/quick_bench.py $1 $2 16 512 0 > gpurun_out/paths_$2_$4.log 2>&1 || { echo "FAILED $cfg"; tail -5 gpurun_out/paths_$2_$4.log; exit 1; }
  echo "== agents $2 B $1 tier $4"; grep "^rep\|resident" gpurun_out/paths_$2_$4.log | tail -2
done
