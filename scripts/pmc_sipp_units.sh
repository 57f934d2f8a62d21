# which unit of a CU the resident SIPP kernel keeps busy (dev tool; run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcsu_$n -- python3 $R/scripts/sipp_probe.py 256 512 1500 > $R/gpurun_out/pmcsu_$n.log 2>&1 || { grep -i "error\|invalid\|not" $R/gpurun_out/pmcsu_$n.log | head -5; continue; }
  python3 - <<PY
import csv, glob
tot = {}
for f in glob.glob("$R/gpurun_out/pmcsu_$n/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "sipp" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
exp = [int(l.split()[1]) for l in open("$R/gpurun_out/pmcsu_$n.log") if l.startswith("total_expansions")][0]
print({k: round(v / exp, 2) for k, v in sorted(tot.items())}, "per expansion;", exp, "expansions")
PY
done
