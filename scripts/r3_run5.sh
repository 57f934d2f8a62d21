set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_run5
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 900 python bench.py --steps 2 --warmup 1 --legs shipped --no-cpu-baseline > $O/bench_sweep.json 2> $O/bench_sweep.err || { tail -5 $O/bench_sweep.err; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_sweep.json") if l.startswith("{")][-1])
print("headline %.3e" % d["value"]); print(json.dumps(d.get("host_threads_sweep"), indent=1))
PY
MRP_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 1 --warmup 0 --instances 8192 --no-cpu-baseline > $O/rehearsal_2ranks_gloo.json 2> $O/rehearsal.err || { tail -5 $O/rehearsal.err; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open("$O/rehearsal_2ranks_gloo.json") if l.startswith("{")][-1])
print(json.dumps(d["by_workload"].get("sharded_conflict_tree"), indent=1))
PY
