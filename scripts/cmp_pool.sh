for mode in static pool; do
  if [ $mode = static ]; then export MRP_HL_STATIC_SPLIT=1; else unset MRP_HL_STATIC_SPLIT; fi
  timeout -k 10 300 python scripts/quick_bench.py 65536 10 16 0 0 > gpurun_out/r02l_${mode}_10.log 2>&1 || exit 1
  timeout -k 10 300 python scripts/quick_bench.py 4096 50 16 0 0 > gpurun_out/r02l_${mode}_50.log 2>&1 || exit 1
  echo "== $mode"; grep "^rep [12]" gpurun_out/r02l_${mode}_10.log gpurun_out/r02l_${mode}_50.log | cut -c1-140
done
