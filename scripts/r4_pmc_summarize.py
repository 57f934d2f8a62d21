"""profiles/<tag>_pmc_summary.json + profiles/hbm_traffic_pmc.json (what bench.py scales roofline.traffic from and quotes
roofline.pipes from) out of scripts/r4_pmc_resident.sh's passes (the resident front + heavy kernels of an ECBS session)
and scripts/r4_pmc_loaded.sh's summary (the batch-mode kernel on a LOADED chip: 3072 resident searches):

    python scripts/r4_pmc_summarize.py gpurun_out/pmcres_<tag> gpurun_out/pmc_loaded_<tag>.json <tag>

Counters of the resident passes are summed over the dispatches of BOTH kernels and divided by the low-level expansions the
profiled bench.py reports; FETCH_SIZE / WRITE_SIZE come in KB (raw values kept; the gfx950 x2 on FETCH_SIZE of
MI355X_MICROARCH.md — calibrated for wide streaming reads — is quoted separately as the upper estimate)."""
import csv
import glob
import json
import os
import sys

bundle, loaded_path, tag = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("mrp_ll_ecbs_front_kernel", "mrp_ll_ecbs_heavy_kernel", "mrp_ll_ecbs_persistent_kernel")
per, disp, regs, exps, line1 = {}, {}, {}, {}, None
for p in range(1, 6):
    try:
        with open(os.path.join(bundle, "ecbs_p%d.log" % p)) as fh:
            b = json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])
    except (OSError, IndexError, ValueError):
        continue
    e = b["value"] * b["ms_per_step"] * b["steps"] / 1000.0
    exps[p] = e
    if line1 is None:
        line1 = {k: b[k] for k in ("value", "ms_per_step", "solved", "instances", "ll_searches", "config", "tiers") if k in b}
    tot = {}
    for f in glob.glob(os.path.join(bundle, "ecbs_p%d" % p, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].split("(")[0]
                if not any(n in k for n in KERNELS):
                    continue
                c = row["Counter_Name"]
                tot[c] = tot.get(c, 0.0) + float(row["Counter_Value"])
                disp.setdefault(c, set()).add(row["Dispatch_Id"])
                regs[k] = dict(vgpr=row["VGPR_Count"], sgpr=row["SGPR_Count"], lds=row["LDS_Block_Size"], grid=row["Grid_Size"])
    for c, v in tot.items():
        per[c] = v / e
g = per.get
out = {"round": tag, "kernels": sorted(regs), "per_expansion": {c: round(v, 4) for c, v in sorted(per.items())},
       "dispatches_per_pass": {c: len(v) for c, v in disp.items()}, "expansions_per_pass": {str(k): v for k, v in exps.items()},
       "kernel_registers": regs}
out["instructions_per_expansion"] = round(sum(g(k, 0.0) for k in ("SQ_INSTS_SALU", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH",
                                                                  "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")), 1)
f_b, w_b = g("FETCH_SIZE", 0.0) * 1024.0, g("WRITE_SIZE", 0.0) * 1024.0
out["fetch_bytes_per_expansion_raw"] = round(f_b, 1)
out["write_bytes_per_expansion_raw"] = round(w_b, 1)
out["bytes_per_expansion_raw"] = round(f_b + w_b, 1)
out["bytes_per_expansion_fetch_x2"] = round(2 * f_b + w_b, 1)
if g("TCC_HIT_sum") is not None:
    out["l2_hit_rate"] = round(g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum", 0.0), 1e-9), 4)
    out["l2_requests_per_expansion"] = round(g("TCC_HIT_sum") + g("TCC_MISS_sum", 0.0), 3)
out["shader_clock_hz"] = 2.4e9
out["sq_units"] = ("the cycle counters of these passes (SQ_WAVE_CYCLES, SQ_WAIT_*) include the time resident workgroups wait for "
                   "the ONE host thread: per-expansion instruction and memory counters only; the pipes below come from the "
                   "loaded-chip pass")
out["bench_line_of_pass_1"] = line1
out["commands"] = ["scripts/r4_pmc_resident.sh %s: five passes of `rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- "
                   "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --legs none --threads 1 --instances 16384`" % tag]
try:
    with open(loaded_path) as fh:
        L = json.load(fh)
    q = L["per_expansion"]
    waves_per_simd = 3
    cyc = 4.0 * q["SQ_WAVE_CYCLES"]  # the SQ cycle counters count in units of four shader cycles
    out["pipes"] = {
        "source": "scripts/r4_pmc_loaded.sh: rocprofv3 --pmc passes over the batch-mode kernel mrp_ll_ecbs_search_kernel with 3072 "
                  "searches resident (12 per CU = 3 waves per SIMD), harvested agents10 searches; committed as profiles/%s_pmc_loaded.json" % tag,
        "instructions_per_expansion": L["instructions_per_expansion"],
        "insts_per_expansion": {k[9:].lower(): q[k] for k in q if k.startswith("SQ_INSTS_")},
        "shader_cycles_per_expansion_per_wave": round(cyc, 0),
        "wave_issuing_fraction": round(q["SQ_ACTIVE_INST_ANY"] / q["SQ_WAVE_CYCLES"], 3),
        "wave_parked_on_waitcnt_fraction": round(q["SQ_WAIT_ANY"] / q["SQ_WAVE_CYCLES"], 3),
        "wave_issue_stalled_fraction": round(q["SQ_WAIT_INST_ANY"] / q["SQ_WAVE_CYCLES"], 3),
        "waves_per_simd": waves_per_simd,
        # MI355X_MICROARCH.md: SIMD-32, a wave64 VALU instruction occupies its SIMD for 2 cycles (4 only for a wave alone)
        "valu_pipe_busy_fraction_of_simd": round(waves_per_simd * q["SQ_INSTS_VALU"] * 2.0 / cyc, 3),
        # one scalar unit per CU (guide), taken at one instruction per cycle, shared by the CU's 12 waves
        "scalar_unit_busy_fraction_of_cu": round(4 * waves_per_simd * (q["SQ_INSTS_SALU"] + q["SQ_INSTS_BRANCH"]) / cyc, 3),
        "lds_instructions_per_expansion": q["SQ_INSTS_LDS"],
        "reading": "per wave: issuing / parked on s_waitcnt (the LDS and memory round trips of dependent heap steps) / stalled at "
                   "issue as given above — latency-bound; per CU with three waves per SIMD: the vector pipes and the one scalar "
                   "unit busy as given above — the scalar unit (uniform control flow, ballots, lane reads) is the fuller pipe, "
                   "none is saturated",
    }
    with open(os.path.join(ROOT, "profiles", "%s_pmc_loaded.json" % tag), "w") as fh:
        json.dump(L, fh, indent=1)
except (OSError, KeyError, ValueError) as e:
    out["pipes"] = None
    print("no loaded-chip summary:", e)
for name in ("%s_pmc_summary.json" % tag, "hbm_traffic_pmc.json"):
    with open(os.path.join(ROOT, "profiles", name), "w") as fh:
        json.dump(out, fh, indent=1)
print({k: out.get(k) for k in ("instructions_per_expansion", "bytes_per_expansion_raw", "l2_hit_rate")})
print(out["pipes"])
