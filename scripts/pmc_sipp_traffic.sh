# HBM traffic of the resident SIPP kernel per expansion (dev tool; run on the GPU box): bash scripts/pmc_sipp_traffic.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-cur}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcst_${c}_$T -- python3 $R/scripts/sipp_bench.py 100 2048 16 0 > $R/gpurun_out/pmcst_${c}_$T.log 2>&1 || { tail -5 $R/gpurun_out/pmcst_${c}_$T.log; exit 1; }
done
python3 - <<PY
import csv, glob, re
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = 0.0
    for f in glob.glob("$R/gpurun_out/pmcst_%s_$T/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            if "sipp" in row["Kernel_Name"] and row["Counter_Name"] == c:
                tot += float(row["Counter_Value"])
    exp = 0.0
    for l in open("$R/gpurun_out/pmcst_%s_$T.log" % c):
        m = re.search(r"LDS tier .* over ([0-9.e+]+) expansions, arena tier .* over ([0-9.e+]+);", l)
        if m: exp += float(m.group(1)) + float(m.group(2))
        m = re.search(r"middle tier .* over ([0-9.e+]+) expansions", l)
        if m: exp += float(m.group(1))
    print(c, "total KB %.4g over %.4g expansions (2 timed passes) -> %.1f bytes per expansion (raw counter, KB units)" % (tot, exp, tot * 1024 / max(exp, 1)))
PY
