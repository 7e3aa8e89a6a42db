#!/bin/bash
# GPU-box script (round 2): kernel-trace stats and PMC counters of the ME chains (12 x 1080p pictures per call) with the shared-plane
# sub-pel kernel.  Separate --pmc passes, no trace domains combined with them.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in batch batch209; do
  timeout -k 5 180 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_$m -o t -- python3 $R/tools/me_picture_probe.py 5 $m > $R/gpurun_out/r02_prof_$m.log 2>&1
done
timeout -k 5 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/r02_pmc_a -o a -- python3 $R/tools/me_picture_probe.py 2 batch209 > $R/gpurun_out/r02_pmc_a.log 2>&1
timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/r02_pmc_b -o b -- python3 $R/tools/me_picture_probe.py 2 batch209 > $R/gpurun_out/r02_pmc_b.log 2>&1
timeout -k 5 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/r02_pmc_c -o c -- python3 $R/tools/me_picture_probe.py 2 batch209 > $R/gpurun_out/r02_pmc_c.log 2>&1
python3 - <<'PY' > $R/gpurun_out/r02_subpel_prof_summary.txt
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for m in ("batch","batch209"):
    for f in glob.glob(f"{R}/gpurun_out/r02_prof_{m}/**/*kernel_stats.csv", recursive=True):
        print("==", m, "kernel stats (name, calls, total ns, avg ns, pct)")
        for row in csv.DictReader(open(f)):
            print("  ", row["Name"].split("(")[0][:60], row["Calls"], row["TotalDurationNs"], row["AverageNs"], row["Percentage"])
for tag in "abc":
    for f in glob.glob(f"{R}/gpurun_out/r02_pmc_{tag}/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]]+=float(row["Counter_Value"]); n[(k,row["Counter_Name"])]+=1
        for k,v in acc.items():
            if 'svthip' in k: print(tag, k[8:], {c: round(x/ n[(k,c)]) for c,x in v.items()})
PY
cat $R/gpurun_out/r02_subpel_prof_summary.txt
