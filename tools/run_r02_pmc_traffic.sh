#!/bin/bash
# GPU-box script (round 2): HBM traffic counters (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) and kernel-trace stats of the bench
# command itself -> gpurun_out/r02_pmc_traffic.json + r02_bench_kernel_stats.csv
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02_pmc_fetch -o f -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02_pmc_fetch.log 2>&1 || { tail -5 $R/gpurun_out/r02_pmc_fetch.log; exit 1; }
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r02_pmc_write -o w -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02_pmc_write.log 2>&1 || { tail -5 $R/gpurun_out/r02_pmc_write.log; exit 1; }
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_bench_trace -o t -- python3 $R/bench.py --steps 20 --no-cpu-baseline > $R/gpurun_out/r02_bench_traced.json 2> $R/gpurun_out/r02_bench_traced.err
python3 - <<'PY'
import csv, glob, collections, os, json
R=os.environ["GRAFT_REPO_ROOT"]
out=collections.defaultdict(dict)
for d in ("r02_pmc_fetch","r02_pmc_write"):
    for f in glob.glob(f"{R}/gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k=(row["Kernel_Name"].split("(")[0], row["Counter_Name"], row.get("Grid_Size","") )
            acc[k].append(float(row["Counter_Value"]))
        for (k,c,g),v in acc.items():
            key=k if not k.startswith("void svthip::") else k[5:]
            out[f"{key}|grid={g}"][c]=sum(v)/len(v); out[f"{key}|grid={g}"]["dispatches_"+c]=len(v)
json.dump(out, open(f"{R}/gpurun_out/r02_pmc_traffic_raw.json","w"), indent=1)
# the bench's dominant launches: fullpel85 over 6120 blocks (grid 6120*256 threads), keep the most frequent grid per kernel
best={}
for k,v in out.items():
    name,grid=k.split("|grid=")
    n=v.get("dispatches_FETCH_SIZE",0)
    if name not in best or n>best[name][0]: best[name]=(n,{kk:vv for kk,vv in v.items()})
json.dump({k:v[1] for k,v in best.items()}, open(f"{R}/gpurun_out/r02_pmc_traffic.json","w"), indent=1)
for k,v in best.items():
    if "svthip" in k: print(k, v[1])
PY
grep -h "svthip" $R/gpurun_out/r02_bench_trace/t_kernel_stats.csv | cut -c1-60,200- | head -20
cut -c1-400 $R/gpurun_out/r02_bench_traced.json
