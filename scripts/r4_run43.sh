set -o pipefail
mkdir -p gpurun_out/r4ba
timeout -k 10 300 python -m pytest tests/test_hl_parity_gpu.py -m gpu -x -q -k "not heavy_tail and not shipped" > gpurun_out/r4ba/tests.log 2>&1 || { tail -20 gpurun_out/r4ba/tests.log; exit 1; }
tail -1 gpurun_out/r4ba/tests.log
for i in 1 2; do
  timeout -k 10 200 python scripts/stream_legs_probe.py 100 16384 3000000 2 > gpurun_out/r4ba/a100_$i.log 2>&1 || { echo failed; exit 1; }
  echo "a100 $i: $(grep '^stream' gpurun_out/r4ba/a100_$i.log)"
done
timeout -k 10 200 python scripts/stream_legs_probe.py 50 65536 400000 3 > gpurun_out/r4ba/a50.log 2>&1 || { echo failed; exit 1; }
echo "a50: $(grep '^stream' gpurun_out/r4ba/a50.log)"
