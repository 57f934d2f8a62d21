for m in 128 64; do
  MRP_HL_STORE_MAX_AGENTS=$m MRP_HL_TIMING=1 MRP_REPS=2 MRP_CAP=2000000 timeout -k 10 250 python scripts/quick_bench.py 2048 100 16 512 0 > gpurun_out/r2l_store100_$m.log 2>&1
  echo "== agents100 store max agents $m"; grep "^rep\|staged" gpurun_out/r2l_store100_$m.log | tail -2; grep "last #0" gpurun_out/r2l_store100_$m.log | sort -k10 -n | tail -4
done
