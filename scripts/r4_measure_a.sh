# Round-4 measurement bundle, part A (run on the MI355X box through gpurun): GPU tests, smoke, the default bench line, the
# kernel-trace stats of the bench, the other shapes (CBS 8x8).  -> gpurun_out/r4a_final/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4a_final
rm -rf $O
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
( time timeout -k 10 1000 python bench.py > $O/bench_line.json 2> $O/bench.err ) 2> $O/bench_time.txt || { tail -5 $O/bench.err; exit 1; }
grep real $O/bench_time.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --legs none > $O/bench_line_under_rocprof.json 2> $O/rocprof.err || { tail -5 $O/rocprof.err; exit 1; }
find $O/prof -name "*kernel_trace.csv" -delete
echo "rocprof stats done"
cd $R
timeout -k 10 500 python scripts/bench_configs.py > $O/bench_configs.log 2>&1 || { tail -5 $O/bench_configs.log; exit 1; }
echo "bench_configs done"
