# PMC passes over the RESIDENT kernels as they are timed (run on the GPU box through gpurun):
#   mrp_ll_ecbs_persistent_kernel under `bench.py --steps 1 --warmup 0 --legs none --threads 1 --instances 16384`
#     (ONE engine = one resident launch of 1792 workgroups: rocprofv3 serialises dispatches while it collects counters,
#     so the sixteen launches of a sixteen-thread step would run one after the other, each waiting for the previous
#     thread's whole share of the batch), and
#   mrp_ll_sipp_persistent_kernel under `scripts/sipp_bench.py 100 8192 16 0` (static split: the sixteen launches do run
#     one after the other here, each over its own instances).
# usage: bash scripts/r3_pmc_resident.sh <tag> [ecbs|sipp|both] ["1 2 5" = passes to run, default all]
#        -> gpurun_out/pmcres_<tag>/<kernel>_p<pass>/ + logs
# Counters are collected in their own runs (--kernel-trace + --pmc only), the program directly after `--`.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-cur}
W=${2:-both}
SEL=${3:-1 2 3 4 5 6 7 8}
O=$R/gpurun_out/pmcres_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_BRANCH"
P2="SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES"
P6="TCC_HIT_sum TCC_MISS_sum"
P7="TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
P8="GRBM_GUI_ACTIVE GRBM_COUNT"
run_pass() {  # kernel-tag pass-name "counters" program args...
  local k=$1 p=$2 c=$3
  shift 3
  timeout -k 10 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${k}_$p -o pmc -- "$@" > $O/${k}_$p.log 2> $O/${k}_$p.err
  local rc=$?
  find $O/${k}_$p -name "*kernel_trace.csv" -delete 2>/dev/null
  echo "$k $p rc=$rc" | tee -a $O/passes.txt
  return 0
}
if [ "$W" = ecbs ] || [ "$W" = both ]; then
  for i in $SEL; do
    eval c=\$P$i
    run_pass ecbs p$i "$c" python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --legs none --threads 1 --instances 16384
  done
fi
if [ "$W" = sipp ] || [ "$W" = both ]; then
  for i in $SEL; do
    eval c=\$P$i
    MRP_NO_CPU=1 run_pass sipp p$i "$c" python3 $R/scripts/sipp_bench.py 100 8192 16 0
  done
fi
echo "pmc resident done"
