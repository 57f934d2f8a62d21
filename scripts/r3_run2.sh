# round-3 GPU run 2: first run of the compact tier (ll_compact.h) on the device
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_run2
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_ll_parity_gpu.py -m gpu -x -q > $O/tests_ll.log 2>&1 || { tail -40 $O/tests_ll.log; exit 1; }
tail -2 $O/tests_ll.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
MRP_REPS=2 timeout -k 10 300 python scripts/quick_bench.py 65536 10 16 > $O/quick10.log 2>&1 || { tail -20 $O/quick10.log; exit 1; }
cat $O/quick10.log | grep -v "^generated\|^solver"
timeout -k 10 300 python bench.py --instances 16384 --steps 3 --warmup 1 --no-cpu-baseline --legs none > $O/B16384.json 2> $O/B16384.err || exit 1
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs shipped > $O/bench_shipped.json 2> $O/bench_shipped.err || exit 1
grep -h -o '"value": [0-9.e+]*' $O/B16384.json $O/bench_shipped.json
echo run2 done
