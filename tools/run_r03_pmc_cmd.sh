#!/bin/bash
# GPU-box script (round 3): arbitrary counter passes over one command, summarised per (kernel, grid).
# Separate --pmc passes, no trace domains combined with them; the program comes directly after `--`.
# usage: PMC_CMD="python3 $GRAFT_REPO_ROOT/bench.py ..." bash tools/run_r03_pmc_cmd.sh <tag> "<pass 1 counters>" "<pass 2 counters>" ...
#        -> gpurun_out/r03_pmcx_<tag>.txt
tag=$1; shift
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
    i=$((i + 1))
    timeout -k 5 300 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/r03_pmcx_${tag}_$i -o p -- $PMC_CMD > $R/gpurun_out/r03_pmcx_${tag}_$i.log 2>&1 \
        || { echo "pass $i ($pass) failed:"; grep -m1 "error code" $R/gpurun_out/r03_pmcx_${tag}_$i.log; }
done
python3 - "$tag" <<'PY' > $R/gpurun_out/r03_pmcx_$tag.txt
import csv, glob, collections, os, sys
R = os.environ["GRAFT_REPO_ROOT"]; tag = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(f"{R}/gpurun_out/r03_pmcx_{tag}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "svthip" not in k: continue
        k = (k.replace("svthip::", "").replace("(anonymous namespace)::", "").split("(")[0], row.get("Grid_Size", ""))
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for k, v in sorted(tot.items()):
    a = {c: x / n[(k, c)] for c, x in v.items()}
    print(k[0], "grid", k[1], "dispatches", max(n[(k, c)] for c in a))
    for c in sorted(a): print(f"    {c:40s} {a[c]:18.1f}")
PY
rm -rf $R/gpurun_out/r03_pmcx_${tag}_[0-9]*
cat $R/gpurun_out/r03_pmcx_$tag.txt | head -150
