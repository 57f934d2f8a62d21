# round-2 measurement bundle: kernel PMC + trace breakdown + scheduler knobs
R=$GRAFT_REPO_ROOT
bash scripts/pmc_ll.sh v3 > gpurun_out/r2e_pmc.log 2>&1; grep -v "^[EW]2026" gpurun_out/r2e_pmc.log | tail -32
MRP_LL_LIB=$R/libmultirobotplanning_amd/lib/libmrp_ll_trace.so timeout -k 10 200 python scripts/prof_ll.py 10 256 > gpurun_out/r2e_trace.log 2>&1; tail -8 gpurun_out/r2e_trace.log
export MRP_REPS=3
for cfg in "65536 0 128" "65536 128 128" "65536 256 128" "65536 512 128" "65536 384 192" "131072 256 128"; do
  set -- $cfg
  MRP_HL_TIMING=1 MRP_HL_RING_DEPTH=$2 MRP_HL_SESSION_WGS=$3 timeout -k 10 200 python scripts/quick_bench.py $1 10 16 512 0 > gpurun_out/ring_$1_$2_$3.log 2>&1 || { echo "FAILED $cfg"; tail -5 gpurun_out/ring_$1_$2_$3.log; exit 1; }
  echo "== B $1 ring depth $2 (0 = default 2 x wgs) wgs/thread $3"; grep "^rep\|resident" gpurun_out/ring_$1_$2_$3.log | tail -2; grep "last #0" gpurun_out/ring_$1_$2_$3.log | tail -2
done
