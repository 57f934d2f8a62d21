# Two host threads (896 resident workgroups per engine) against sixteen: the same batches, several processes each
# (round 3 saw the two-thread case fall into a slow state in some processes and not in others).
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
run() { echo "== $*"; env "$@" MRP_REPS=2 timeout -k 5 120 python scripts/quick_bench.py $B 10 $T $S 2>&1 | grep "^rep\|kernel tiers\|busy fraction\|host thread"; }
B=65536 T=2 S=1024 run A=0
B=65536 T=2 S=1024 run A=1
B=65536 T=2 S=1024 run A=2
B=32768 T=2 S=1024 run A=3
B=65536 T=4 S=512 run A=4
B=65536 T=16 S=512 run A=5
