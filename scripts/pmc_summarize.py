"""Per-expansion summary of the rocprofv3 --pmc passes of scripts/pmc_ll.sh (reads gpurun_out/pmc*/ ... counter_collection.csv).

usage: python scripts/pmc_summarize.py <expansions per launch> <dir> [<dir> ...]
Sums every counter over the dispatches of the low-level search kernels (name contains "mrp_ll") and divides by the
number of such dispatches and by the expansions one launch processes."""
import csv
import glob
import json
import os
import sys

exp = float(sys.argv[1])
tot, launches, regs = {}, {}, {}
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "mrp_ll" not in row["Kernel_Name"]:
                    continue
                c = row["Counter_Name"]
                tot[c] = tot.get(c, 0.0) + float(row["Counter_Value"])
                key = (c, row["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    launches[c] = launches.get(c, 0) + 1
                regs[row["Kernel_Name"]] = dict(vgpr=row["VGPR_Count"], sgpr=row["SGPR_Count"], lds=row["LDS_Block_Size"])
out = {c: round(tot[c] / max(launches[c], 1) / exp, 2) for c in sorted(tot)}
ipe = sum(out.get(k, 0) for k in ("SQ_INSTS_SALU", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD",
                                   "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"))
print(json.dumps(dict(per_expansion=out, instructions_per_expansion=round(ipe, 1), launches=launches.get("SQ_INSTS_SALU", 0),
                      kernels=regs), indent=1))
