# PMC passes over the batch-mode kernel on harvested ECBS low-level searches (dev tool; run on the GPU box)
# usage: bash scripts/pmc_ll.sh [tag]   -> gpurun_out/pmc1_<tag>, pmc2_<tag> (+ .log)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-cur}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pmc1_$T -- python3 $R/scripts/prof_ll.py 10 256 > $R/gpurun_out/pmc1_$T.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/pmc2_$T -- python3 $R/scripts/prof_ll.py 10 256 > $R/gpurun_out/pmc2_$T.log 2>&1 || exit 1
E=$(grep "^jobs" $R/gpurun_out/pmc1_$T.log | awk '{print $4}')
python3 $R/scripts/pmc_summarize.py $E $R/gpurun_out/pmc1_$T $R/gpurun_out/pmc2_$T > $R/gpurun_out/pmc_summary_$T.json
cat $R/gpurun_out/pmc_summary_$T.json
