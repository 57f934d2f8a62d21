# occupancy sweep of the resident kernel: LDS tier geometry (nodes rows path-bytes) x workgroups per host thread
N=${1:-32768}
for cfg in "512 64 4096 64" "256 48 2048 128" "256 48 2048 96" "384 56 2048 80"; do
  set -- $cfg
  MRP_LDS_NODES=$1 MRP_LL_LDS_ROWS=$2 MRP_LL_LDS_PATHS=$3 MRP_HL_SESSION_WGS=$4 timeout -k 10 300 python scripts/quick_bench.py $N 10 16 $4 0 > gpurun_out/occ_$1_$2_$3_$4.log 2>&1 || exit 1
  echo "== nodes $1 rows $2 paths $3 wgs/thread $4"; grep "^rep\|resident" gpurun_out/occ_$1_$2_$3_$4.log | tail -4
done
