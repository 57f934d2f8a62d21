# A/B of two builds of the libraries on one box (dev tool): build an earlier commit's csrc into libmultirobotplanning_amd/lib_old/ first (git show <rev>:... + hipcc); lib/ is the current build; static split for both
export MRP_HL_STATIC_SPLIT=1
for rep in 1 2; do
  for which in old new new1; do
    unset MRP_HL_LIB MRP_HL_TICKETS
    if [ $which = old ]; then export MRP_HL_LIB=$PWD/libmultirobotplanning_amd/lib_old/libmrp_hl.so; fi
    if [ $which = new1 ]; then export MRP_HL_TICKETS=1; fi
    timeout -k 10 300 python scripts/quick_bench.py 65536 10 16 0 0 > gpurun_out/ab_${which}_$rep.log 2>&1 || exit 1
    echo "$which $rep: $(grep '^rep [12]' gpurun_out/ab_${which}_$rep.log | sed -E 's/.*wall ([0-9.]+)s.*/\1/' | tr '\n' ' ')  busy $(grep resident gpurun_out/ab_${which}_$rep.log | tail -1 | sed -E 's/.*busy ([0-9.]+) s.*/\1/')"
  done
done
