#!/bin/bash
# round 4, run 42: deep conflict trees balanced over the workers at admission (MRP_HL_DEEP_AT, 0 = off): the agents100 leg
# (host-bound per worker) and the streamed headline, on and off
set -o pipefail
mkdir -p gpurun_out/r4ay
run() {  # name legs steps env...
  n=$1; legs=$2; st=$3; shift 3
  env "$@" timeout -k 10 400 python bench.py --steps $st --warmup 1 --legs $legs --no-cpu-baseline --sync-steps 0 > gpurun_out/r4ay/$n.json 2> gpurun_out/r4ay/$n.err || { echo "failed $n"; tail -5 gpurun_out/r4ay/$n.err; exit 1; }
  python - "$n" <<'P'
import json, sys
d = json.loads([l for l in open("gpurun_out/r4ay/%s.json" % sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
b = d.get("by_workload", {})
print("== %s: headline %.4g (%.1f ms/step)" % (sys.argv[1], d["value"], d["ms_per_step"]) + "".join(
    " | %s %.4g in %.2f s" % (k, v["value"], v["seconds"]) for k, v in b.items() if "seconds" in v and v.get("seconds")), flush=True)
P
}
run off agents100,agents50 6 MRP_HL_DEEP_AT=0
run on agents100,agents50 6 A=1
run off2 agents100 2 MRP_HL_DEEP_AT=0
run on2 agents100 2 A=1
