R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo; lscpu | grep "NUMA node[0-9]"
run() { echo "== $*"; env "$@" MRP_REPS=3 timeout -k 5 200 python scripts/quick_bench.py 262144 10 16 512 2>&1 | grep "^rep [12]\|host thread\|busy fraction" | tail -3; }
run A=0
run MRP_HL_PIN=0,1
run MRP_HL_PIN=64,1
run MRP_HL_PIN=0,2
run MRP_HL_PIN=32,1
run MRP_HL_PIN=96,1
