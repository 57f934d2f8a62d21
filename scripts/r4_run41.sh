#!/bin/bash
# round 4, run 41: fifteen searches per CU in the front tier (run 37's "slim" variant: narrow geometry of 3 groups, front
# kernel held to 128 VGPRs) under the STREAMED headline, where the device and not the tail of a step is the bound
set -o pipefail
mkdir -p gpurun_out/r4ax
L=$PWD/libmultirobotplanning_amd/lib
LD_PRELOAD=$L/libmrp_ll_slim.so timeout -k 10 600 python -m pytest tests/test_hl_parity_gpu.py -m gpu -x -q -k "stream or synthetic" > gpurun_out/r4ax/pytest_slim.log 2>&1; rc=$?; echo "pytest (slim) rc=$rc $(tail -1 gpurun_out/r4ax/pytest_slim.log)"
[ $rc -eq 0 ] || exit $rc
run() {  # name preload env...
  n=$1; pl=$2; shift 2
  ( [ -n "$pl" ] && export LD_PRELOAD=$pl; env "$@" timeout -k 10 300 python bench.py --steps 6 --warmup 1 --legs none --no-cpu-baseline --sync-steps 0 ) > gpurun_out/r4ax/$n.json 2> gpurun_out/r4ax/$n.err || { echo "failed $n"; tail -5 gpurun_out/r4ax/$n.err; exit 1; }
  python - "$n" <<'P'
import json, sys
d = json.loads([l for l in open("gpurun_out/r4ax/%s.json" % sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
t = d["tiers"]
print("== %s: %.4g exp/s, %.1f ms/step, front busy %.3f (%.2f us/exp over %.3g), heavy %d busy %.3f (%.2f us/exp over %.3g), handed over %d" % (
    sys.argv[1], d["value"], d["ms_per_step"], t["front_workgroups_busy_fraction"], t["front_us_per_expansion"], t["front_expansions"],
    t["heavy_workgroups"], t["heavy_workgroups_busy_fraction"], t["beyond_front_us_per_expansion"], t["beyond_front_expansions"], t["searches_handed_over"]), flush=True)
P
}
run default "" A=1
run slim $L/libmrp_ll_slim.so A=1
run slim_heavy224 $L/libmrp_ll_slim.so MRP_HL_HEAVY_WGS=224
run slim_heavy256 $L/libmrp_ll_slim.so MRP_HL_HEAVY_WGS=256
