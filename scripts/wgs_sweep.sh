R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
run() { echo "== $*"; env "$@" MRP_REPS=3 timeout -k 5 200 python scripts/quick_bench.py $B 10 $T $S 2>&1 | grep "^rep 2\|kernel tiers\|busy fraction\|host thread\|whole job" | tail -5; }
B=262144 T=16 S=512 run A=0
B=262144 T=16 S=512 run MRP_HL_ROOT_CHAIN=0
B=65536 T=2 S=1024 run A=0
B=65536 T=16 S=512 run A=0
