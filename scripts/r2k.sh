for ps in 262144 0; do
  MRP_HL_PATH_SLOTS=$ps MRP_REPS=2 MRP_CAP=2000000 timeout -k 10 250 python scripts/quick_bench.py 2048 100 16 512 0 > gpurun_out/r2k_store100_$ps.log 2>&1
  echo "== agents100 path store slots $ps"; grep "^rep\|staged" gpurun_out/r2k_store100_$ps.log | tail -2
  MRP_HL_PATH_SLOTS=$ps MRP_REPS=2 MRP_CAP=400000 timeout -k 10 250 python scripts/quick_bench.py 8192 50 16 512 0 > gpurun_out/r2k_store50_$ps.log 2>&1
  echo "== agents50 path store slots $ps"; grep "^rep\|staged" gpurun_out/r2k_store50_$ps.log | tail -2
  MRP_HL_PATH_SLOTS=$ps MRP_REPS=3 timeout -k 10 250 python scripts/quick_bench.py 65536 10 16 512 4096 > gpurun_out/r2k_store10_$ps.log 2>&1
  echo "== agents10 path store slots $ps"; grep "^rep\|staged\|cpu oracle" gpurun_out/r2k_store10_$ps.log | tail -3
done
