#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4s
(time timeout -k 10 900 python bench.py --steps 1 --warmup 0 --instances 4096 --no-cpu-baseline --legs shipped_heavy_tail) > gpurun_out/r4s/ex36.log 2> gpurun_out/r4s/ex36.err; echo "ex36 rc=$?"; tail -4 gpurun_out/r4s/ex36.err
python -c "
import json
d=json.loads(open('gpurun_out/r4s/ex36.log').read().strip().splitlines()[-1])
print(d['by_workload'].get('shipped_heavy_tail'))"
timeout -k 10 300 python scripts/rehearse_sharded.py > gpurun_out/r4s/rehearsal_world1.json 2> gpurun_out/r4s/rehearsal_world1.err; echo "world1 rc=$?"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 scripts/rehearse_sharded.py > gpurun_out/r4s/rehearsal_world2.json 2> gpurun_out/r4s/rehearsal_world2.err; echo "world2 rc=$?"
tail -1 gpurun_out/r4s/rehearsal_world1.json | cut -c1-1500; tail -1 gpurun_out/r4s/rehearsal_world2.json | cut -c1-1500
