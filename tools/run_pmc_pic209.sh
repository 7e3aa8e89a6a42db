#!/bin/bash
# GPU-box script: PMC counters of the whole-picture ME kernels (batched B pictures).
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_pic209_a -o a -- python3 $R/tools/me_picture_probe.py 2 batch209 > $R/gpurun_out/pmc_pic209_a.log 2>&1
timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_pic209_b -o b -- python3 $R/tools/me_picture_probe.py 2 batch209 > $R/gpurun_out/pmc_pic209_b.log 2>&1
timeout -k 5 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_pic209_c -o c -- python3 $R/tools/me_picture_probe.py 2 batch209 > $R/gpurun_out/pmc_pic209_c.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for tag in "abc":
    for f in glob.glob(f"{R}/gpurun_out/pmc_pic209_{tag}/*counter_collection.csv"):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]]+=float(row["Counter_Value"]); n[(k,row["Counter_Name"])]+=1
        for k,v in acc.items():
            if 'svthip' in k: print(tag, k[8:], {c: round(x/ n[(k,c)]) for c,x in v.items()})
PY
