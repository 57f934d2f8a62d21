# Round-end measurement bundle (run on the MI355X box through gpurun); writes everything under gpurun_out/final/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 600 python bench.py > $O/bench_line.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_line_under_rocprof.json 2> $O/rocprof.err || { tail -5 $O/rocprof.err; exit 1; }
find $O/prof -name "*kernel_trace.csv" -delete
find $O/prof -name "*_stats.csv" | head
echo "rocprof stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --instances 16384 --no-cpu-baseline > $O/pmc_$c.json 2> $O/pmc_$c.err || { tail -5 $O/pmc_$c.err; exit 1; }
  find $O/pmc_$c -name "*kernel_trace.csv" -delete
done
echo "pmc done"
bash $R/scripts/pmc_ll.sh > $O/pmc_ll.log 2>&1 || { tail -5 $O/pmc_ll.log; exit 1; }
echo "pmc_ll done"
cd $R
timeout -k 10 300 python bench.py --agents 20 --instances 32768 --steps 2 --warmup 1 --cpu-sample 1024 > $O/bench_agents20.json 2> $O/bench_agents20.err || exit 1
timeout -k 10 300 python bench.py --agents 50 --instances 4096 --steps 2 --warmup 1 --cpu-sample 128 > $O/bench_agents50.json 2> $O/bench_agents50.err || exit 1
echo "agents20/50 done"
timeout -k 10 400 python scripts/bench_configs.py > $O/bench_configs.log 2>&1 || { tail -5 $O/bench_configs.log; exit 1; }
timeout -k 10 300 python scripts/sipp_bench.py 50 4096 16 16 > $O/sipp50.log 2>&1 || exit 1
timeout -k 10 300 python scripts/sipp_bench.py 100 2048 16 8 > $O/sipp100.log 2>&1 || exit 1
echo "all done"
