#!/bin/bash
# round 4, run 28: conflict-tree look-ahead width on the tail-bound legs (all 1000 shipped inputs; agents100)
set -o pipefail
mkdir -p gpurun_out/r4af
for k in 2 4 8 16; do
  MRP_HL_SPEC=$k timeout -k 10 400 python bench.py --steps 1 --warmup 0 --instances 16384 --no-cpu-baseline --legs shipped,agents100 > gpurun_out/r4af/spec$k.json 2> gpurun_out/r4af/spec$k.err || { echo failed $k; tail -5 gpurun_out/r4af/spec$k.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4af/spec$k.json") if l.startswith("{")][-1])
w=d["by_workload"]
print("spec $k: shipped %.3f s (%d mismatches), agents100 %.3f s" % (w["shipped"]["seconds"], w["shipped"]["parity_mismatches_vs_golden"], w["agents100"]["seconds"]))
PY
done
