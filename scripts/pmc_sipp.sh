# PMC passes over the resident SIPP kernel (dev tool; run on the GPU box): bash scripts/pmc_sipp.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-cur}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pmcs1_$T -- python3 $R/scripts/sipp_probe.py 16 256 1500 > $R/gpurun_out/pmcs1_$T.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/pmcs2_$T -- python3 $R/scripts/sipp_probe.py 16 256 1500 > $R/gpurun_out/pmcs2_$T.log 2>&1 || exit 1
E=$(grep "^total_expansions" $R/gpurun_out/pmcs1_$T.log | awk '{print $2}')
python3 $R/scripts/pmc_summarize.py $E $R/gpurun_out/pmcs1_$T $R/gpurun_out/pmcs2_$T > $R/gpurun_out/pmc_sipp_summary_$T.json
cat $R/gpurun_out/pmc_sipp_summary_$T.json
