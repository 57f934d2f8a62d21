# Round-end measurement bundle (run on the MI355X box through gpurun); writes everything under gpurun_out/final/
# usage: bash scripts/final_measure.sh [tag]      (tag names the files that go to profiles/, e.g. r02)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-r02}
O=$R/gpurun_out/final
rm -rf $O
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 900 python bench.py > $O/bench_line.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs none > $O/bench_line_under_rocprof.json 2> $O/rocprof.err || { tail -5 $O/rocprof.err; exit 1; }
find $O/prof -name "*kernel_trace.csv" -delete
find $O/prof -name "*_stats.csv" | head
echo "rocprof stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --instances 16384 --no-cpu-baseline --legs none > $O/pmc_$c.json 2> $O/pmc_$c.err || { tail -5 $O/pmc_$c.err; exit 1; }
  find $O/pmc_$c -name "*kernel_trace.csv" -delete
done
echo "pmc done"
bash $R/scripts/pmc_ll.sh $T > $O/pmc_ll.log 2>&1 || { tail -5 $O/pmc_ll.log; exit 1; }
cp $R/gpurun_out/pmc_summary_$T.json $O/
echo "pmc_ll done"
cd $R
MRP_LL_LIB=$R/libmultirobotplanning_amd/lib/libmrp_ll_trace.so timeout -k 10 200 python scripts/prof_ll.py 10 256 > $O/trace_breakdown.log 2>&1 || { tail -5 $O/trace_breakdown.log; exit 1; }
python scripts/make_pmc_summary.py $O $T > $O/pmc_summary.log 2>&1 || { cat $O/pmc_summary.log; exit 1; }
timeout -k 10 400 python scripts/bench_configs.py > $O/bench_configs.log 2>&1 || { tail -5 $O/bench_configs.log; exit 1; }
timeout -k 10 300 python scripts/sipp_bench.py 50 8192 16 512 > $O/sipp50.log 2>&1 || exit 1
timeout -k 10 300 python scripts/sipp_bench.py 100 8192 16 512 > $O/sipp100.log 2>&1 || exit 1
timeout -k 10 300 python scripts/sipp_bench.py 100 2048 16 512 > $O/sipp100_2048.log 2>&1 || exit 1
timeout -k 10 300 python scripts/sipp_bench.py 200 4096 16 512 > $O/sipp200.log 2>&1 || exit 1
bash $R/scripts/pmc_sipp.sh $T > $O/pmc_sipp.log 2>&1 || { tail -5 $O/pmc_sipp.log; exit 1; }
cp $R/gpurun_out/pmc_sipp_summary_$T.json $O/
cd $R
for a in 10 50; do
  MRP_REPS=2 MRP_CAP=$([ $a = 10 ] && echo 50000 || echo 400000) timeout -k 10 300 python scripts/quick_bench.py $([ $a = 10 ] && echo 131072 || echo 65536) $a 16 > $O/quick_agents$a.log 2>&1 || exit 1
done
timeout -k 10 300 python scripts/spec_probe.py > $O/spec_probe.log 2>&1 || exit 1
for b in 16384 65536 262144; do
  timeout -k 10 300 python bench.py --instances $b --steps 2 --warmup 1 --no-cpu-baseline --legs none > $O/bench_B$b.json 2> $O/bench_B$b.err || exit 1
done
MRP_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 1 --warmup 0 --instances 8192 --no-cpu-baseline > $O/rehearsal_2ranks_gloo.json 2> $O/rehearsal.err || { tail -5 $O/rehearsal.err; exit 1; }
echo "all done"
