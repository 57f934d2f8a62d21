#!/bin/bash
# round 4, run 38: few worker threads — is the step host-bound, or bound by the job slots (= resident workgroups) an engine has?
set -o pipefail
mkdir -p gpurun_out/r4au
run() {  # name threads slots
  MRP_CAP=50000 MRP_REPS=3 timeout -k 10 500 python scripts/quick_bench.py 262144 10 $2 $3 > gpurun_out/r4au/$1.log 2>&1 || { echo "failed $1"; tail -5 gpurun_out/r4au/$1.log; exit 1; }
  echo "== $1: $(grep '^rep' gpurun_out/r4au/$1.log | awk '{print $4}' | tr '\n' ' ') busy $(grep 'busy fraction' gpurun_out/r4au/$1.log | tail -1 | awk '{print $NF}') | $(grep 'resident workgroups' gpurun_out/r4au/$1.log | tail -1 | cut -c1-80)"
}
run t2_s512 2 512
run t2_s1536 2 1536
run t4_s512 4 512
run t4_s768 4 768
run t4_s1024 4 1024
run t8_s512 8 512
