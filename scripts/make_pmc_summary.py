"""profiles/<tag>_pmc_summary.json (= profiles/hbm_traffic_pmc.json, which bench.py scales roofline.traffic / issue_bound
from) out of one measurement bundle:  python scripts/make_pmc_summary.py <bundle dir> <tag>

  <bundle>/pmc_FETCH_SIZE, <bundle>/pmc_WRITE_SIZE : rocprofv3 --pmc passes over bench.py (+ its JSON line next to them)
  <bundle>/pmc_summary_<tag>.json                   : scripts/pmc_ll.sh (SQ passes over harvested searches)
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; raw values are kept (the gfx950 x2 on FETCH_SIZE is calibrated
for wide streaming reads only — MI355X_MICROARCH.md, HBM — and is quoted separately as the upper estimate)."""
import csv
import glob
import json
import os
import sys

bundle, tag = sys.argv[1], sys.argv[2]
out = {"round": tag, "hbm_passes": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, launches = 0.0, set()
    for f in glob.glob(os.path.join(bundle, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "mrp_ll" in row["Kernel_Name"] and row["Counter_Name"] == c:
                    tot += float(row["Counter_Value"])
                    launches.add((f, row["Dispatch_Id"]))
    with open(os.path.join(bundle, "pmc_%s.json" % c)) as fh:
        line = [l for l in fh.read().splitlines() if l.startswith("{")][-1]
    b = json.loads(line)
    exp = b["value"] * b["ms_per_step"] * b["steps"] / 1000.0
    out["hbm_passes"][c] = {"launches": len(launches), "sum_kb": tot, "per_launch_kb": tot / max(len(launches), 1),
                            "bench_expansions": exp, "bytes_per_expansion_raw": tot * 1024.0 / max(exp, 1.0)}
f_b = out["hbm_passes"]["FETCH_SIZE"]["bytes_per_expansion_raw"]
w_b = out["hbm_passes"]["WRITE_SIZE"]["bytes_per_expansion_raw"]
out["fetch_bytes_per_expansion_raw"] = round(f_b, 1)
out["write_bytes_per_expansion_raw"] = round(w_b, 1)
out["bytes_per_expansion_raw"] = round(f_b + w_b, 1)
out["bytes_per_expansion_fetch_x2"] = round(2 * f_b + w_b, 1)
with open(os.path.join(bundle, "pmc_summary_%s.json" % tag)) as fh:
    sq = json.load(fh)
out["instructions_per_expansion"] = sq["instructions_per_expansion"]
out["sq_counters_per_expansion"] = sq["per_expansion"]
out["kernel_registers"] = sq["kernels"]
pe = sq["per_expansion"]
if pe.get("SQ_WAVE_CYCLES"):
    out["issue_active_fraction"] = round(pe.get("SQ_ACTIVE_INST_ANY", 0) / pe["SQ_WAVE_CYCLES"], 3)
    out["wait_fraction"] = round(pe.get("SQ_WAIT_ANY", 0) / pe["SQ_WAVE_CYCLES"], 3)
out["sq_units"] = "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles (x4 = shader cycles)"
out["shader_clock_hz"] = 2.2e9
out["commands"] = [
    "rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 1 --warmup 0 --instances 16384 --no-cpu-baseline --legs none",
    "the same with --pmc WRITE_SIZE (separate pass)",
    "scripts/pmc_ll.sh: two SQ_* passes over scripts/prof_ll.py 10 256 (batch-mode ECBS kernel on harvested searches)"]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in ("%s_pmc_summary.json" % tag, "hbm_traffic_pmc.json"):
    with open(os.path.join(root, "profiles", name), "w") as fh:
        json.dump(out, fh, indent=1)
print(json.dumps({k: out[k] for k in ("bytes_per_expansion_raw", "instructions_per_expansion")}))
