R=$GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_ll_parity_gpu.py tests/test_hl_parity_gpu.py -m gpu -x -q > gpurun_out/r2m_pytest.log 2>&1 || { tail -20 gpurun_out/r2m_pytest.log; exit 1; }
tail -2 gpurun_out/r2m_pytest.log
export MRP_REPS=3
for v in regs lds regs lds; do
  if [ $v = lds ]; then export LD_PRELOAD=$R/libmultirobotplanning_amd/lib/libmrp_ll_ldsparams.so; else unset LD_PRELOAD; fi
  timeout -k 10 200 python scripts/quick_bench.py 65536 10 16 512 0 > gpurun_out/r2m_$v.log 2>&1 || { tail -5 gpurun_out/r2m_$v.log; exit 1; }
  echo "== launch params in $v"; grep "^rep\|resident" gpurun_out/r2m_$v.log | tail -4
done
unset LD_PRELOAD
for m in 128 64; do
  MRP_HL_STORE_MAX_AGENTS=$m MRP_REPS=2 MRP_CAP=2000000 timeout -k 10 250 python scripts/quick_bench.py 2048 100 16 512 0 > gpurun_out/r2m_store100_$m.log 2>&1
  echo "== agents100 store max agents $m"; grep "^rep\|staged" gpurun_out/r2m_store100_$m.log | tail -2
done
bash scripts/pmc_ll.sh v4 > gpurun_out/r2m_pmc.log 2>&1; grep -A3 "instructions_per_expansion\|SQ_INSTS_SALU\|SQ_INSTS_VALU\|SQ_WAVE_CYCLES" gpurun_out/pmc_summary_v4.json | head -20
