# admission limit / batch size sweep of the session driver (agents10): where does the step time go, throughput or tail?
export MRP_REPS=2
for cfg in "65536 256" "65536 512" "65536 1536" "65536 4096" "131072 1536" "131072 8192"; do
  set -- $cfg
  MRP_HL_TIMING=1 MRP_HL_ACTIVE_LIMIT=$2 timeout -k 10 200 python scripts/quick_bench.py $1 10 16 512 0 > gpurun_out/sched_$1_$2.log 2>&1 || { echo "FAILED $cfg"; tail -5 gpurun_out/sched_$1_$2.log; exit 1; }
  echo "== B $1 active limit/thread $2"; grep "^rep\|resident" gpurun_out/sched_$1_$2.log | tail -2; grep "last #0" gpurun_out/sched_$1_$2.log | sort -t' ' -k9 -n | tail -3
done
