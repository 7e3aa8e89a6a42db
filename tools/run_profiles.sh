#!/bin/bash
# GPU-box script: the round's measured artefacts -- bench line (with CPU baseline), rocprofv3 kernel stats of the same
# command, and a separate PMC pass (FETCH_SIZE / WRITE_SIZE) for the HBM traffic of the dominant kernel.
set -e
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python3 $R/bench.py > $R/gpurun_out/bench_full.json 2> $R/gpurun_out/bench_full.err
cat $R/gpurun_out/bench_full.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_traffic -o t -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_traffic.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os, json
R=os.environ["GRAFT_REPO_ROOT"]
for f in glob.glob(f"{R}/gpurun_out/prof_bench/*kernel_stats.csv"):
    print(open(f).read()[:1200])
out={}
for f in glob.glob(f"{R}/gpurun_out/pmc_traffic/*counter_collection.csv"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]]+=float(row["Counter_Value"]); n[(k,row["Counter_Name"])]+=1
    for k,v in acc.items():
        out[k]={c: x/n[(k,c)] for c,x in v.items()}
        out[k]["dispatches"]=n[(k,"FETCH_SIZE")]
print(json.dumps(out, indent=1))
json.dump(out, open(f"{R}/gpurun_out/pmc_traffic_summary.json","w"), indent=1)
PY
