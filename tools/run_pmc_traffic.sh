#!/bin/bash
# GPU-box script: HBM traffic counters of the ME kernels, one derived counter per pass (both together exceed what the
# hardware can collect at once).
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1 || { tail -5 $R/gpurun_out/pmc_fetch.log; exit 1; }
timeout -k 5 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1 || { tail -5 $R/gpurun_out/pmc_write.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, os, json
R=os.environ["GRAFT_REPO_ROOT"]
out=collections.defaultdict(dict)
for d in ("pmc_fetch","pmc_write"):
    for f in glob.glob(f"{R}/gpurun_out/{d}/*counter_collection.csv"):
        acc=collections.defaultdict(float); n=collections.Counter()
        for row in csv.DictReader(open(f)):
            k=(row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            acc[k]+=float(row["Counter_Value"]); n[k]+=1
        for (k,c),v in acc.items():
            out[k][c]=v/n[(k,c)]; out[k]["dispatches_"+c]=n[(k,c)]
print(json.dumps(out, indent=1))
json.dump(out, open(f"{R}/gpurun_out/pmc_traffic_summary.json","w"), indent=1)
PY
