#!/bin/bash
# GPU-box script: HBM traffic counters of the transform / quantisation kernels (FETCH_SIZE and WRITE_SIZE in separate passes).
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
SZ=${1:-16x16,32x32}
cd /tmp && export TMPDIR=/tmp
timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_tq_fetch -o f -- python3 $R/tools/tq_probe.py --sizes $SZ --iters 2 > $R/gpurun_out/pmc_tq_fetch.log 2>&1 || { tail -5 $R/gpurun_out/pmc_tq_fetch.log; exit 1; }
timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_tq_write -o w -- python3 $R/tools/tq_probe.py --sizes $SZ --iters 2 > $R/gpurun_out/pmc_tq_write.log 2>&1 || { tail -5 $R/gpurun_out/pmc_tq_write.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, os, json, re
R=os.environ["GRAFT_REPO_ROOT"]
out=collections.defaultdict(dict)
for d in ("pmc_tq_fetch","pmc_tq_write"):
    for f in glob.glob(f"{R}/gpurun_out/{d}/*counter_collection.csv"):
        acc=collections.defaultdict(float); n=collections.Counter()
        for row in csv.DictReader(open(f)):
            name=row["Kernel_Name"]
            m=re.search(r"(fwd_txfm2d_kernel|inv_txfm2d_add_kernel|encode_tu_kernel|quantize_b_batch_kernel)(<[^>]*>)?", name)
            if not m: continue
            k=(m.group(0), row["Counter_Name"])
            acc[k]+=float(row["Counter_Value"]); n[k]+=1
        for (k,c),v in acc.items():
            out[k][c+"_KB_per_launch"]=v/n[(k,c)]; out[k]["dispatches_"+c]=n[(k,c)]
print(json.dumps(out, indent=1))
json.dump(out, open(f"{R}/gpurun_out/pmc_traffic_tq.json","w"), indent=1)
PY
