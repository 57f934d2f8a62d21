# PMC passes over the RESIDENT kernel of an ECBS session under `bench.py --steps 1 --warmup 0 --legs none --threads 1
# --instances 16384` with MRP_HL_HEAVY_WGS=0: ONE engine and ONE launch (mrp_ll_ecbs_persistent_kernel: the narrow LDS tier
# + the arena tier), because rocprofv3 serialises dispatches while it collects counters — the front / heavy pair of the
# timed configuration cannot run under it (the front launch would wait for the heavy one to end; the engine reports that
# as an error).  The narrow tier's code is the same in both.  Counters in their own runs (--kernel-trace + --pmc only).
# usage (GPU box): bash scripts/r4_pmc_resident.sh <tag>  -> gpurun_out/pmcres_<tag>/ecbs_p<pass>/ + logs
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-r04}
O=$R/gpurun_out/pmcres_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MRP_HL_HEAVY_WGS=0
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_BRANCH"
P2="SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="TCC_HIT_sum TCC_MISS_sum"
for i in 1 2 3 4 5; do
  eval c=\$P$i
  timeout -k 10 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/ecbs_p$i -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --legs none --threads 1 --instances 16384 > $O/ecbs_p$i.log 2> $O/ecbs_p$i.err
  echo "ecbs p$i rc=$?" | tee -a $O/passes.txt
  find $O/ecbs_p$i -name "*kernel_trace.csv" -delete 2>/dev/null
done
echo "pmc resident done"
