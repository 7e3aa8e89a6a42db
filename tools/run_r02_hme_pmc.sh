#!/bin/bash
# GPU-box script: where does the search-centre kernel's time go?  SQ counters of tools/hme_probe.py (separate --pmc passes).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 5 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/hme_pmc_$tag -o p -- python3 $R/tools/hme_probe.py > $R/gpurun_out/hme_pmc_$tag.log 2>&1 || { tail -3 $R/gpurun_out/hme_pmc_$tag.log; }
done
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{R}/gpurun_out/hme_pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "hme_center" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]][int(row["Dispatch_Id"])].append(float(row["Counter_Value"]))
# the probe launches 23 x 4 configs; dispatches come in order: all levels first
for c,v in sorted(acc.items()):
    ids=sorted(v)
    per=[sum(v[i]) for i in ids]
    n=len(per)//4
    print(c, " all-levels %.3e  L0+L1 %.3e  L0 %.3e  centre %.3e" % tuple(sum(per[k*n:(k+1)*n])/max(1,n) for k in range(4)))
PY
