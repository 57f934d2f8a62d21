#!/bin/bash
# round 4, run 29: CBS 8x8 (config 3; one resident kernel per engine, tiny searches): 8 engines with co-workers against 16 engines
set -o pipefail
mkdir -p gpurun_out/r4ag
for e in 8 16; do
  MRP_HL_MAX_ENGINES=$e timeout -k 10 400 python scripts/bench_configs.py > gpurun_out/r4ag/engines$e.log 2>&1 || { echo failed $e; tail -5 gpurun_out/r4ag/engines$e.log; exit 1; }
  echo "== engines $e"; grep -o '"cbs_8x8_agents[0-9]*": {"instances": [0-9]*, "solved": [0-9]*, "gpu_exp_per_s": [0-9.e+]*' gpurun_out/r4ag/engines$e.log
done
