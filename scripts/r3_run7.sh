set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_run7
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for b in 262144 65536 16384; do
MRP_HL_TIMING=1 MRP_REPS=3 timeout -k 5 300 python scripts/quick_bench.py $b 10 16 512 > $O/q$b.log 2>&1 || { tail -5 $O/q$b.log; exit 1; }
grep "^rep\|kernel tiers\|busy fraction" $O/q$b.log | tail -4; grep "active wgs" $O/q$b.log | tail -1 | cut -c1-200
done
