R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
run() { echo "== $*"; env "$@" MRP_NO_CPU=1 timeout -k 5 200 python scripts/sipp_bench.py 100 8192 16 0 2>&1 | grep "^rep 1\|SIPP kernel\|middle tier" | tail -3; }
run A=0
run MRP_LL_SIPP_TABLES_UNCACHED=1
MRP_LL_SIPP_TABLES_UNCACHED=1 timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "sipp" 2>&1 | tail -3
