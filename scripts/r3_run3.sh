set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_run3
mkdir -p $O
cd $R
timeout -k 10 300 python scripts/occupancy_latency.py 10 256 1,64,256,512,1024,1792 > $O/occ10.log 2>&1 || { tail -20 $O/occ10.log; exit 1; }
cat $O/occ10.log
bash scripts/pmc_ll.sh r3a > $O/pmc_ll.log 2>&1 || { tail -5 $O/pmc_ll.log; exit 1; }
cat gpurun_out/pmc_summary_r3a.json
