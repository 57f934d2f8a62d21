# Round-3 measurement bundle, part B (GPU box): PMC passes over the resident SIPP kernel, SQ passes over the batch-mode
# ECBS kernel with every workgroup busy, the compact tier's phase profile, latency against occupancy, batch sizes,
# CBS 8x8 (config 3), thread counts, the two-rank rehearsal of the sharded conflict tree.  -> gpurun_out/r3b/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3b
rm -rf $O
mkdir -p $O
cd $R
bash $R/scripts/r3_pmc_resident.sh r03 sipp > $O/pmc_sipp.log 2>&1 || { tail -5 $O/pmc_sipp.log; exit 1; }
tail -2 $O/pmc_sipp.log
bash $R/scripts/pmc_ll.sh r03 > $O/pmc_ll.log 2>&1 || { tail -5 $O/pmc_ll.log; exit 1; }
cp $R/gpurun_out/pmc_summary_r03.json $O/pmc_batch_kernel_summary.json
echo "pmc_ll done"
cd $R
MRP_LL_LIB=$R/libmultirobotplanning_amd/lib/libmrp_ll_ctprof.so timeout -k 10 200 python scripts/ct_phase_profile.py 10 64 1 > $O/ct_phase_profile.txt 2>&1 || { tail -5 $O/ct_phase_profile.txt; exit 1; }
timeout -k 10 300 python scripts/occupancy_latency.py 10 256 1,256,1024,1792,3072 > $O/occupancy_latency.txt 2>&1 || { tail -5 $O/occupancy_latency.txt; exit 1; }
echo "profiles done"
for b in 16384 65536 262144 524288; do
  timeout -k 10 300 python bench.py --instances $b --steps 2 --warmup 1 --no-cpu-baseline --legs none > $O/bench_B$b.json 2> $O/bench_B$b.err || exit 1
done
echo "batch sizes done"
timeout -k 10 400 python scripts/bench_configs.py > $O/bench_configs.log 2>&1 || { tail -5 $O/bench_configs.log; exit 1; }
MRP_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 1 --warmup 0 --instances 8192 --no-cpu-baseline > $O/rehearsal_2ranks_gloo.json 2> $O/rehearsal.err || { tail -5 $O/rehearsal.err; exit 1; }
echo "all done"
