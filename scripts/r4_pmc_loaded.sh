# PMC passes over the batch-mode ECBS kernel on a LOADED chip (3072 resident searches, harvested agents10 searches): the
# per-pipe figures bench.py's roofline.pipes quotes.  Run on the GPU box: bash scripts/r4_pmc_loaded.sh [tag]
# -> gpurun_out/pmcL{1,2,3}_<tag>/ (+ .log) and gpurun_out/pmc_loaded_<tag>.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r04}
N=${2:-1536}
run() {  # pass-name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcL${name}_$T -- python3 $R/scripts/pmc_jobs.py $N 3072 > $R/gpurun_out/pmcL${name}_$T.log 2>&1 || { echo "pass $name failed"; tail -5 $R/gpurun_out/pmcL${name}_$T.log; exit 1; }
}
run 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS
run 2 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU
run 3 SQ_BUSY_CU_CYCLES SQ_WAVES SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT
E=$(grep "^jobs" $R/gpurun_out/pmcL1_$T.log | awk '{print $4}')
python3 $R/scripts/pmc_summarize.py $E $R/gpurun_out/pmcL1_$T $R/gpurun_out/pmcL2_$T $R/gpurun_out/pmcL3_$T > $R/gpurun_out/pmc_loaded_$T.json
grep "^rep" $R/gpurun_out/pmcL1_$T.log
cat $R/gpurun_out/pmc_loaded_$T.json
