# Round-4 measurement bundle, part B (GPU box): batch-size sweep, SIPP counters, the two-rank
# rehearsal of bench.py on one GPU over gloo, kernel resources.  -> gpurun_out/r4b_final/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4b_final
rm -rf $O
mkdir -p $O
cd $R
for b in 16384 65536 262144 524288; do
  timeout -k 10 300 python bench.py --instances $b --steps 3 --warmup 1 --no-cpu-baseline --legs none > $O/bench_B$b.json 2> $O/bench_B$b.err || { tail -5 $O/bench_B$b.err; exit 1; }
  echo "B=$b done"
done
MRP_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --instances 65536 --no-cpu-baseline --legs none > $O/rehearsal_2ranks_gloo.json 2> $O/rehearsal.err || { tail -5 $O/rehearsal.err; exit 1; }
echo "rehearsal done"
bash $R/scripts/r3_pmc_resident.sh r04s sipp "1 3 4 6" > $O/pmc_sipp.log 2>&1 || { tail -5 $O/pmc_sipp.log; exit 1; }
tail -2 $O/pmc_sipp.log
