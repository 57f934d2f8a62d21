# LDS tier geometry x resident workgroups sweep of the ECBS resident kernel (quick_bench: agents10, N instances)
N=${1:-65536}
export MRP_REPS=2
for cfg in "400,48,2048 128 0" "400,48,2048 192 0" "384,43,2048 192 0" "320,40,1536 224 0" "256,36,1536 256 0"; do
  set -- $cfg
  MRP_HL_TIER=$1 MRP_HL_SESSION_WGS=$2 MRP_LL_EXTRA_HBM_WGS=$3 timeout -k 10 200 python scripts/quick_bench.py $N 10 16 512 0 > gpurun_out/tier_$1_$2_$3.log 2>&1 || { echo "FAILED $cfg"; tail -5 gpurun_out/tier_$1_$2_$3.log; exit 1; }
  echo "== tier $1 wgs/thread $2 extra-hbm $3"; grep "^rep\|resident" gpurun_out/tier_$1_$2_$3.log | tail -2
done
# the same with a second resident launch of LDS-less workgroups per engine (needs a hardware queue per stream)
for cfg in "400,48,2048 128 64" "400,48,2048 192 64"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=40 MRP_HL_TIER=$1 MRP_HL_SESSION_WGS=$2 MRP_LL_EXTRA_HBM_WGS=$3 timeout -k 10 200 python scripts/quick_bench.py $N 10 16 512 0 > gpurun_out/tier_$1_$2_$3.log 2>&1 || { echo "FAILED $cfg"; tail -5 gpurun_out/tier_$1_$2_$3.log; exit 1; }
  echo "== tier $1 wgs/thread $2 extra-hbm $3"; grep "^rep\|resident" gpurun_out/tier_$1_$2_$3.log | tail -2
done
