# does the size of the per-engine arena allocation matter? (dev tool)
for rep in 1 2; do
  for slots in 0 128 256; do
    timeout -k 10 300 python scripts/quick_bench.py 65536 10 16 $slots 0 > gpurun_out/slots_${slots}_$rep.log 2>&1 || exit 1
    echo "slots $slots rep $rep: $(grep '^rep [12]' gpurun_out/slots_${slots}_$rep.log | sed -E 's/.*wall ([0-9.]+)s.*/\1/' | tr '\n' ' ')  busy $(grep resident gpurun_out/slots_${slots}_$rep.log | tail -1 | sed -E 's/.*busy ([0-9.]+) s.*/\1/')"
  done
done
