# A/B of the resident loop's per-job cache fences (GPU box): the product build against -DMRP_LL_SESSION_FENCES
# (python -c "from libmultirobotplanning_amd import _build; _build.build_variant('fences', ['-DMRP_LL_SESSION_FENCES'])").
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
run() { echo "== $*"; env "$@" MRP_REPS=3 timeout -k 5 200 python scripts/quick_bench.py 262144 10 16 512 2>&1 | grep "^rep 2\|kernel tiers\|busy fraction" | tail -3; }
run BUILD=product
run LD_PRELOAD=$R/libmultirobotplanning_amd/lib/libmrp_ll_fences.so
run BUILD=product MRP_HL_ROOT_CHAIN=0
run LD_PRELOAD=$R/libmultirobotplanning_amd/lib/libmrp_ll_fences.so MRP_HL_ROOT_CHAIN=0
