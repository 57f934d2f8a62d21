set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_run4
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 900 python bench.py --steps 2 --warmup 1 --legs shipped > $O/bench_shipped.json 2> $O/bench_shipped.err || { tail -5 $O/bench_shipped.err; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_shipped.json") if l.startswith("{")][-1])
print("headline %.3e  cpu1 %.3e allcores %.3e" % (d["value"], d["cpu_baseline"]["value"], d["cpu_baseline_all_cores"]["value"]))
print(json.dumps(d["by_workload"]["shipped"], indent=1))
PY
