# Round-4 measurement bundle, part C (GPU box): the streamed headline (bench.py hands its K timed steps to the solver as one
# stream, mrp_hl_solver_solve_stream).  GPU tests, smoke, the bench as the driver runs it, the same with one call per step,
# kernel-trace stats, batch-size sweep, two-rank rehearsal.  -> gpurun_out/r4c_final/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4c_final
rm -rf $O
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
( time timeout -k 10 1000 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err ) 2> $O/bench_time.txt || { tail -5 $O/bench.err; exit 1; }
grep real $O/bench_time.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-stream --no-cpu-baseline --legs none > $O/bench_line_one_call_per_step.json 2> $O/bench_ns.err || { tail -5 $O/bench_ns.err; exit 1; }
echo "one call per step done"
for b in 16384 65536 262144 524288; do
  timeout -k 10 300 python bench.py --instances $b --steps 6 --warmup 1 --no-cpu-baseline --legs none > $O/bench_B$b.json 2> $O/bench_B$b.err || { tail -5 $O/bench_B$b.err; exit 1; }
  echo "B=$b done"
done
MRP_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --instances 65536 --no-cpu-baseline --legs none > $O/rehearsal_2ranks_gloo.json 2> $O/rehearsal.err || { tail -5 $O/rehearsal.err; exit 1; }
echo "rehearsal done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --legs none --sync-steps 0 > $O/bench_line_under_rocprof.json 2> $O/rocprof.err || { tail -5 $O/rocprof.err; exit 1; }
find $O/prof -name "*kernel_trace.csv" -delete
echo "rocprof stats done"
