# prioritized SIPP: is it the 16 host threads? (needs a hardware queue per worker stream)
for t in 16 24 32; do
  GPU_MAX_HW_QUEUES=40 timeout -k 10 200 python scripts/sipp_bench.py 100 2048 $t 8 > gpurun_out/r2n_sipp100_t$t.log 2>&1
  echo "== SIPP agents100, $t host threads"; grep "^rep\|engine totals" gpurun_out/r2n_sipp100_t$t.log | tail -2
done
# the headline with 24 / 32 host threads
for t in 24 32; do
  GPU_MAX_HW_QUEUES=40 MRP_REPS=2 timeout -k 10 200 python scripts/quick_bench.py 131072 10 $t 512 0 > gpurun_out/r2n_quick_t$t.log 2>&1
  echo "== agents10 B=131072, $t host threads"; grep "^rep\|resident" gpurun_out/r2n_quick_t$t.log | tail -2
done
MRP_REPS=2 timeout -k 10 200 python scripts/quick_bench.py 131072 10 16 512 0 > gpurun_out/r2n_quick_t16.log 2>&1
echo "== agents10 B=131072, 16 host threads"; grep "^rep\|resident" gpurun_out/r2n_quick_t16.log | tail -2
