#!/bin/bash
# GPU-box script (round 3): SQ counters per kernel for the launches of tools/kernel_times.py (6120 superblocks per launch).
# Separate --pmc passes (8 SQ slots each), no trace domains combined with them; the program comes directly after `--`.
# usage: bash tools/run_r03_pmc_kernels.sh <tag> <what...>     -> gpurun_out/r03_pmc_<tag>.txt
tag=$1; shift
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { timeout -k 5 150 rocprofv3 --pmc "${@:2}" --output-format csv -d $R/gpurun_out/r03_pmc_${tag}_$1 -o p -- python3 $R/tools/kernel_times.py $WHAT > $R/gpurun_out/r03_pmc_${tag}_$1.log 2>&1 || { tail -5 $R/gpurun_out/r03_pmc_${tag}_$1.log; exit 1; }; }
WHAT="$*"
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_WAVES
run c GRBM_GUI_ACTIVE FETCH_SIZE
run d WRITE_SIZE
python3 - "$tag" <<'PY' > $R/gpurun_out/r03_pmc_$tag.txt
import csv, glob, collections, os, sys
R=os.environ["GRAFT_REPO_ROOT"]; tag=sys.argv[1]
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for p in "abcd":
    for f in glob.glob(f"{R}/gpurun_out/r03_pmc_{tag}_{p}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"].split("(")[0]
            if "svthip" not in k: continue
            tot[k][row["Counter_Name"]]+=float(row["Counter_Value"]); n[(k,row["Counter_Name"])]+=1
for k,v in tot.items():
    a={c: x/n[(k,c)] for c,x in v.items()}
    print(k.replace("svthip::",""), "dispatches", n[(k,"SQ_INSTS_VALU")])
    for c in sorted(a): print(f"    {c:24s} {a[c]:16.0f}")
    if "SQ_BUSY_CYCLES" in a and "SQ_ACTIVE_INST_VALU" in a:
        print(f"    VALU-active quad-cycles x4 / (BUSY_CYCLES/8 XCD x 1024 SIMD-share): valu_busy ~ {4*a['SQ_ACTIVE_INST_VALU']/(a['SQ_WAVE_CYCLES']*4/ max(1,a.get('SQ_WAVES',1))* 0 + 1):.0f} (raw)")
    if "SQ_INSTS_LDS" in a and a["SQ_INSTS_LDS"]>0: print(f"    bank conflict cycles per LDS instruction: {a['SQ_LDS_BANK_CONFLICT']/a['SQ_INSTS_LDS']:.2f}")
    if "FETCH_SIZE" in a: print(f"    HBM-side traffic per launch: 2 x FETCH_SIZE + WRITE_SIZE = {(2*a['FETCH_SIZE']+a.get('WRITE_SIZE',0))/1e3:.1f} MB (FETCH/WRITE_SIZE in KB)")
PY
cat $R/gpurun_out/r03_pmc_$tag.txt
