set -o pipefail
mkdir -p gpurun_out/r4az
for i in 1 2 3; do
for m in off on; do
  if [ $m = off ]; then export MRP_HL_DEEP_AT=0; else unset MRP_HL_DEEP_AT; fi
  timeout -k 10 200 python scripts/stream_legs_probe.py 100 16384 3000000 2 > gpurun_out/r4az/${m}_$i.log 2>&1 || { echo failed; exit 1; }
  echo "$m $i: $(grep '^stream' gpurun_out/r4az/${m}_$i.log)"
done
done
