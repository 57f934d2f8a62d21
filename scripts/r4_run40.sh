#!/bin/bash
# round 4, run 40: the streamed headline (bench.py's default since mrp_hl_solver_solve_stream) against the knobs that were
# tuned for one call per step: heavy workgroups, device-queue depth
set -o pipefail
mkdir -p gpurun_out/r4aw
run() {  # name env...
  n=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 6 --warmup 1 --legs none --no-cpu-baseline --sync-steps 0 > gpurun_out/r4aw/$n.json 2> gpurun_out/r4aw/$n.err || { echo "failed $n"; tail -5 gpurun_out/r4aw/$n.err; exit 1; }
  python - "$n" <<'P'
import json, sys
d = json.loads(open("gpurun_out/r4aw/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
t = d["tiers"]
print("== %s: %.4g exp/s, %.1f ms/step, front busy %.3f (%.2f us/exp), heavy %d busy %.3f (%.2f us/exp)" % (
    sys.argv[1], d["value"], d["ms_per_step"], t["front_workgroups_busy_fraction"], t["front_us_per_expansion"],
    t["heavy_workgroups"], t["heavy_workgroups_busy_fraction"], t["beyond_front_us_per_expansion"]), flush=True)
P
}
run base A=1
run heavy96 MRP_HL_HEAVY_WGS=96
run heavy128 MRP_HL_HEAVY_WGS=128
run heavy192 MRP_HL_HEAVY_WGS=192
run ring3x MRP_HL_RING_DEPTH=1100
run ring15x MRP_HL_RING_DEPTH=560
run base2 A=1
