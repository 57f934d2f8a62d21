R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
run() { echo "== $*"; env "$@" MRP_REPS=3 timeout -k 5 120 python scripts/quick_bench.py $B 10 $T $S 2>&1 | grep "^rep [12]\|kernel tiers\|busy fraction\|host thread" | tail -5; }
B=65536 T=2 S=1024 run A=0
B=65536 T=2 S=1024 run A=1
B=65536 T=4 S=512 run A=2
