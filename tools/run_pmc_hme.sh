#!/bin/bash
# GPU-box script: PMC counters of the ME kernels during a short bench run (separate passes, counters only).
set -e
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_a -o a -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_b -o b -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --fused > $R/gpurun_out/pmc_b.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc_c -o c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_c.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc_d -o d -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_d.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for tag in "abc":
    for f in glob.glob(f"{R}/gpurun_out/pmc_{tag}/*counter_collection.csv"):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"].split("(")[0][-40:]
            acc[k][row["Counter_Name"]]+=float(row["Counter_Value"]); n[(k,row["Counter_Name"])]+=1
        for k,v in acc.items():
            if 'rocclr' in k: continue
            print(tag, k, {c: round(x/ n[(k,c)]) for c,x in v.items()})
for f in glob.glob(f"{R}/gpurun_out/pmc_d/*kernel_stats.csv"):
    print(open(f).read()[:1500])
PY
