# round-3 GPU run 1: tests after the ABI / poll fixes, ring-poll A/B, first PMC passes on the resident kernels
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_run1
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do
  timeout -k 10 300 python bench.py --instances 16384 --steps 3 --warmup 1 --no-cpu-baseline --legs none > $O/B16384_relaxed_$rep.json 2> $O/B16384_relaxed_$rep.err || exit 1
  LD_PRELOAD=$R/libmultirobotplanning_amd/lib/libmrp_ll_pollacq.so timeout -k 10 300 python bench.py --instances 16384 --steps 3 --warmup 1 --no-cpu-baseline --legs none > $O/B16384_acquire_$rep.json 2> $O/B16384_acquire_$rep.err || exit 1
done
grep -h -o '"value": [0-9.e+]*' $O/B16384_*.json
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs shipped > $O/bench_shipped.json 2> $O/bench_shipped.err || exit 1
bash scripts/r3_pmc_resident.sh base ecbs "1 2" && bash scripts/r3_pmc_resident.sh base sipp "1 5 6 7"
echo run1 done
