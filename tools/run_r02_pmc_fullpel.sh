#!/bin/bash
# GPU-box script (round 2, late): LDS / wait counters of the two full-pel kernels after the reduce-scatter and exchange-layout changes
# -> gpurun_out/r02_pmc_fullpel.txt  (separate --pmc passes, tools/fullpel_probe.py launches both kernels over 6120 SBs)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_fpa -o a -- python3 $R/tools/fullpel_probe.py 6120 > $R/gpurun_out/pmc_fpa.log 2>&1 || exit 1
timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_fpb -o b -- python3 $R/tools/fullpel_probe.py 6120 > $R/gpurun_out/pmc_fpb.log 2>&1 || exit 1
timeout -k 5 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc_fpc -o c -- python3 $R/tools/fullpel_probe.py 6120 > $R/gpurun_out/pmc_fpc.log 2>&1 || exit 1
python3 - > $R/gpurun_out/r02_pmc_fullpel.txt <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
print("full-pel kernels over 6120 SBs (tools/fullpel_probe.py 6120), average per launch, rocprofv3 --pmc in three passes")
for tag in "abc":
    for f in glob.glob(f"{R}/gpurun_out/pmc_fp{tag}/*counter_collection.csv"):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]]+=float(row["Counter_Value"]); n[(k,row["Counter_Name"])]+=1
        for k,v in acc.items():
            if 'svthip' in k: print(tag, k[8:], {c: round(x/ n[(k,c)]) for c,x in v.items()})
PY
cat $R/gpurun_out/r02_pmc_fullpel.txt
