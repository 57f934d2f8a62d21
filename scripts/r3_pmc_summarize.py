"""profiles/<tag>_pmc_summary.json, profiles/<tag>_pmc_sipp_summary.json and profiles/hbm_traffic_pmc.json (what bench.py
scales roofline.traffic / issue_bound from) out of the passes of scripts/r3_pmc_resident.sh:

    python scripts/r3_pmc_summarize.py gpurun_out/pmcres_<tag> <tag>

Every counter is summed over the dispatches of the RESIDENT kernel of the pass and divided by the low-level expansions
the profiled program reports for itself (bench.py's JSON line / sipp_bench.py's rep lines).  FETCH_SIZE / WRITE_SIZE are
reported in KB by rocprofv3; raw values are kept, the gfx950 x2 on FETCH_SIZE (MI355X_MICROARCH.md, HBM: calibrated for
wide streaming reads) is quoted separately as the upper estimate."""
import csv
import glob
import json
import os
import re
import sys

bundle, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(kind, kernel_substr):
    tot, disp, regs = {}, {}, {}
    for p in range(1, 9):
        for f in glob.glob(os.path.join(bundle, "%s_p%d" % (kind, p), "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if kernel_substr not in row["Kernel_Name"]:
                        continue
                    c = row["Counter_Name"]
                    tot[c] = tot.get(c, 0.0) + float(row["Counter_Value"])
                    disp.setdefault(c, set()).add(row["Dispatch_Id"])
                    regs[row["Kernel_Name"].split("(")[0]] = dict(vgpr=row["VGPR_Count"], accum_vgpr=row["Accum_VGPR_Count"],
                                                                   sgpr=row["SGPR_Count"], lds=row["LDS_Block_Size"],
                                                                   grid=row["Grid_Size"], workgroup=row["Workgroup_Size"])
    return tot, {c: len(v) for c, v in disp.items()}, regs


def ecbs_expansions(p):
    with open(os.path.join(bundle, "ecbs_p%d.log" % p)) as fh:
        line = [l for l in fh.read().splitlines() if l.startswith("{")][-1]
    b = json.loads(line)
    return b["value"] * b["ms_per_step"] * b["steps"] / 1000.0, b


def sipp_expansions(p):
    tot = 0.0
    n_inst = None
    with open(os.path.join(bundle, "sipp_p%d.log" % p)) as fh:
        for l in fh:
            m = re.search(r"rep \d+: (\d+) instances x \d+ agents: wall ([0-9.]+) s, ([0-9.e+]+) exp/s", l)
            if m:
                n_inst = int(m.group(1))
                tot += float(m.group(2)) * float(m.group(3))
    reps = 2
    return tot * (1.0 + 64.0 / (n_inst * reps)) if n_inst else 0.0  # + the 64-instance warm-up call of the script


def summary(kind, kernel_substr, exp_of):
    tot, disp, regs = counters(kind, kernel_substr)
    exps = {}
    per = {}
    passes = {}
    for p in range(1, 9):
        try:
            e = exp_of(p)
            exps[p] = e[0] if isinstance(e, tuple) else e
        except (OSError, IndexError, ValueError):
            continue
    # which pass a counter came from: re-read the per-pass files (cheap) to pair a counter with ITS run's expansions
    for p in exps:
        for f in glob.glob(os.path.join(bundle, "%s_p%d" % (kind, p), "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if kernel_substr in row["Kernel_Name"]:
                        passes[row["Counter_Name"]] = p
    for c, v in tot.items():
        e = exps.get(passes.get(c, 0), 0.0)
        if e > 0:
            per[c] = v / e
    out = {"round": tag, "kernel": kernel_substr, "per_expansion": {c: round(v, 4) for c, v in sorted(per.items())},
           "dispatches_per_pass": disp, "expansions_per_pass": {str(k): v for k, v in exps.items()}, "kernel_registers": regs}
    g = per.get
    ipe = sum(g(k, 0.0) for k in ("SQ_INSTS_SALU", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD",
                                   "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"))
    out["instructions_per_expansion"] = round(ipe, 1)
    f_b, w_b = g("FETCH_SIZE", 0.0) * 1024.0, g("WRITE_SIZE", 0.0) * 1024.0
    out["fetch_bytes_per_expansion_raw"] = round(f_b, 1)
    out["write_bytes_per_expansion_raw"] = round(w_b, 1)
    out["bytes_per_expansion_raw"] = round(f_b + w_b, 1)
    out["bytes_per_expansion_fetch_x2"] = round(2 * f_b + w_b, 1)
    if g("TCC_HIT_sum") is not None:
        out["l2_hit_rate"] = round(g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum", 0.0), 1e-9), 4)
        out["l2_requests_per_expansion"] = round(g("TCC_HIT_sum") + g("TCC_MISS_sum", 0.0), 3)
    if g("SQ_INSTS_LDS"):
        # SQ_LDS_BANK_CONFLICT counts cycles an LDS instruction spent stalled on bank conflicts; SQ_LDS_IDX_ACTIVE the
        # cycles LDS index instructions were active (both in the unit rocprofv3 reports for SQ_* cycle counters)
        out["lds_bank_conflict_cycles_per_lds_instruction"] = round(g("SQ_LDS_BANK_CONFLICT", 0.0) / g("SQ_INSTS_LDS"), 3)
        if g("SQ_LDS_IDX_ACTIVE"):
            out["lds_bank_conflict_fraction_of_lds_active"] = round(g("SQ_LDS_BANK_CONFLICT", 0.0) / g("SQ_LDS_IDX_ACTIVE"), 4)
    if g("GRBM_GUI_ACTIVE") and g("TA_TA_BUSY_sum") is not None:
        out["ta_busy_note"] = "TA_TA_BUSY_sum is summed over the texture-addresser instances; TA_BUSY_avr is their average"
    out["shader_clock_hz"] = 2.4e9
    out["sq_units"] = ("SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* include the time resident workgroups "
                       "wait for the host (one host thread cannot keep 1792 workgroups busy): use the instruction and "
                       "memory counters per expansion, not the cycle ratios, from these passes")
    return out


e = summary("ecbs", "mrp_ll_ecbs_persistent_kernel", ecbs_expansions)
try:
    e["bench_line_of_pass_1"] = {k: v for k, v in ecbs_expansions(1)[1].items()
                                 if k in ("value", "ms_per_step", "solved", "instances", "ll_searches", "config")}
except (OSError, IndexError, ValueError):
    pass
e["commands"] = ["scripts/r3_pmc_resident.sh %s ecbs: eight passes of `rocprofv3 --kernel-trace --pmc <counters> --output-format csv "
                 "-- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --legs none --threads 1 --instances 16384`" % tag]
s = summary("sipp", "mrp_ll_sipp_persistent_kernel", sipp_expansions)
s["commands"] = ["scripts/r3_pmc_resident.sh %s sipp: eight passes of `rocprofv3 --kernel-trace --pmc <counters> --output-format csv "
                 "-- python3 scripts/sipp_bench.py 100 8192 16 0` (MRP_NO_CPU=1)" % tag]
for name, obj in (("%s_pmc_summary.json" % tag, e), ("hbm_traffic_pmc.json", e), ("%s_pmc_sipp_summary.json" % tag, s)):
    if obj["per_expansion"]:
        with open(os.path.join(ROOT, "profiles", name), "w") as fh:
            json.dump(obj, fh, indent=1)
for o in (e, s):
    print(o["kernel"], {k: o.get(k) for k in ("instructions_per_expansion", "bytes_per_expansion_raw", "l2_hit_rate",
                                                "lds_bank_conflict_cycles_per_lds_instruction")})
