# PMC passes over the prioritized-SIPP bench under full load (dev tool; run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-cur}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmcb1_$T -- python3 $R/scripts/sipp_bench.py 100 2048 16 0 > $R/gpurun_out/pmcb1_$T.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $R/gpurun_out/pmcb2_$T -- python3 $R/scripts/sipp_bench.py 100 2048 16 0 > $R/gpurun_out/pmcb2_$T.log 2>&1 || { tail -5 $R/gpurun_out/pmcb2_$T.log; exit 1; }
grep "^rep" $R/gpurun_out/pmcb1_$T.log $R/gpurun_out/pmcb2_$T.log
python3 - <<PY
import csv, glob
for d in ("$R/gpurun_out/pmcb1_$T", "$R/gpurun_out/pmcb2_$T"):
    tot = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "sipp" in row["Kernel_Name"]:
                tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    print(d.split("/")[-1], {k: "%.4g" % v for k, v in sorted(tot.items())})
PY
