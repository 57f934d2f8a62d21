R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 900 python bench.py > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err || tail -5 gpurun_out/bench_now.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_now.json") if l.startswith("{")][-1])
print('value %.4g ms/step %.1f'%(d['value'],d['ms_per_step']))
print(json.dumps({k:{'v':'%.4g'%v['value'],'r':round(v['vs_16_threads'],3)} for k,v in d['host_threads_sweep'].items()}))
for k,v in d['by_workload'].items():
    print(k, {kk:(('%.4g'%vv) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('value','instances','seconds','solved','capped','parity_mismatches_vs_golden','vs_cpu_port_1core','vs_cpu_port_all_cores')})
PY
