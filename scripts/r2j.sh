for w in 64 96 128; do
  MRP_HL_SIPP_WGS=$w timeout -k 10 300 python scripts/sipp_bench.py 100 2048 16 8 > gpurun_out/r2j_sipp100_$w.log 2>&1
  echo "== SIPP agents100 wgs/thread $w"; grep "^rep\|engine totals" gpurun_out/r2j_sipp100_$w.log | tail -2
done
for ps in 262144 0; do
  MRP_HL_PATH_SLOTS=$ps MRP_REPS=2 MRP_CAP=2000000 timeout -k 10 250 python scripts/quick_bench.py 2048 100 16 512 0 > gpurun_out/r2j_store100_$ps.log 2>&1
  echo "== agents100 path store slots $ps"; grep "^rep\|staged" gpurun_out/r2j_store100_$ps.log | tail -2
  MRP_HL_PATH_SLOTS=$ps MRP_REPS=2 MRP_CAP=400000 timeout -k 10 250 python scripts/quick_bench.py 8192 50 16 512 0 > gpurun_out/r2j_store50_$ps.log 2>&1
  echo "== agents50 path store slots $ps"; grep "^rep\|staged" gpurun_out/r2j_store50_$ps.log | tail -2
done
