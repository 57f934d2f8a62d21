"""Copy the judged summaries of one measurement bundle (gpurun_out/final, written by scripts/final_measure.sh) into
profiles/ under a round tag:  python scripts/collect_profiles.py r02"""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "final")
dst = os.path.join(root, "profiles")


def last_json_line(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def dump(name, obj):
    with open(os.path.join(dst, "%s_%s" % (tag, name)), "w") as f:
        json.dump(obj, f, indent=1)


dump("bench_line.json", last_json_line(os.path.join(src, "bench_line.json")))
dump("bench_line_under_rocprof.json", last_json_line(os.path.join(src, "bench_line_under_rocprof.json")))
for f in glob.glob(os.path.join(src, "prof", "*_stats.csv")):
    shutil.copy(f, os.path.join(dst, "%s_%s" % (tag, os.path.basename(f))))
sweep = {}
for f in sorted(glob.glob(os.path.join(src, "bench_B*.json"))):
    b = last_json_line(f)
    sweep[os.path.basename(f)[7:-5]] = {k: b[k] for k in ("value", "ms_per_step", "instances_per_s", "solved", "instances")}
dump("batch_size_sweep.json", sweep)
other = {}
with open(os.path.join(src, "bench_configs.log")) as f:
    for l in f.read().splitlines():
        if l.startswith("{"):
            other.update(json.loads(l))
for name in ("sipp50", "sipp100", "sipp200"):
    p = os.path.join(src, name + ".log")
    if os.path.exists(p):
        with open(p) as f:
            other["prioritized_" + name + "_log_tail"] = [l.strip() for l in f.read().splitlines() if l.startswith(("rep", "cpu port", "   engine"))][-3:]
dump("other_shapes.json", other)
for name in ("spec_probe.log", "trace_breakdown.log"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        with open(p) as f:
            keep = [l for l in f.read().splitlines() if l and not l.startswith(("[W", "/opt"))]
        with open(os.path.join(dst, "%s_%s" % (tag, name.replace(".log", ".txt"))), "w") as f:
            f.write("\n".join(keep[-40:]) + "\n")
p = os.path.join(src, "rehearsal_2ranks_gloo.json")
if os.path.exists(p):
    dump("rehearsal_2ranks_one_gpu_gloo.json", last_json_line(p))
print(sorted(os.listdir(dst)))
