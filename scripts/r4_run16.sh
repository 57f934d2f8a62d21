#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4r
MRP_HL_TIMING=1 MRP_REPS=2 timeout -k 10 300 python scripts/quick_bench.py 262144 10 2 1536 > gpurun_out/r4r/w2_timing.log 2>&1
grep "rep 1\|host thread-seconds" gpurun_out/r4r/w2_timing.log | tail -2
grep "host ms" gpurun_out/r4r/w2_timing.log | tail -2
(time timeout -k 10 900 python bench.py --steps 1 --warmup 0 --instances 4096 --no-cpu-baseline --legs shipped_heavy_tail) > gpurun_out/r4r/ex36.log 2> gpurun_out/r4r/ex36.err; echo "ex36 rc=$?"
tail -3 gpurun_out/r4r/ex36.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r4r/ex36.log").read().strip().splitlines()[-1])
print(d["by_workload"].get("shipped_heavy_tail"))
PY
