#!/bin/bash
# round 4, run 18: lane-mask predicates (wave_dev.h) against bool predicates — parity tests, per-expansion time, instruction counts
set -o pipefail
mkdir -p gpurun_out/r4u
timeout -k 10 600 python -m pytest tests/test_ll_parity_gpu.py -m gpu -x -q > gpurun_out/r4u/pytest_ll.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4u/pytest_ll.log
[ $rc -eq 0 ] || exit $rc
for v in mask bool; do
  if [ $v = bool ]; then export MRP_LL_LIB=$PWD/libmultirobotplanning_amd/lib/libmrp_ll_boolpred.so; else unset MRP_LL_LIB; fi
  timeout -k 10 300 python scripts/pmc_jobs.py 768 3072 > gpurun_out/r4u/jobs_$v.log 2>&1 || { echo "jobs $v failed"; tail -3 gpurun_out/r4u/jobs_$v.log; exit 1; }
  echo "== $v"; grep "^rep" gpurun_out/r4u/jobs_$v.log
done
