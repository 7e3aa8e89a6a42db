#!/bin/bash
# GPU-box script (round 3): counters of EVERY kernel the bench command launches -> gpurun_out/r03_pmc_traffic.json (copied to profiles/).
# Separate rocprofv3 --pmc passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2: they cannot share a pass), no trace domains combined with
# them, the program directly after `--`; then the kernel-trace stats of the same command.
#   per (kernel, grid, workgroup): FETCH_SIZE / WRITE_SIZE (KB per launch), GRBM_GUI_ACTIVE (sum over the 8 XCDs),
#   SQ_ACTIVE_INST_VALU (quad-cycles), SQ_INSTS_VALU, SQ_INSTS_LDS, SQ_LDS_BANK_CONFLICT, SQ_WAVES
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
run() { timeout -k 5 400 rocprofv3 --pmc "${@:2}" --output-format csv -d $R/gpurun_out/r03_pmc_bench_$1 -o p -- $CMD > $R/gpurun_out/r03_pmc_bench_$1.log 2>&1 || { tail -5 $R/gpurun_out/r03_pmc_bench_$1.log; exit 1; }; }
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
run sq SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_bench_trace -o t -- python3 $R/bench.py --steps 20 --no-cpu-baseline > $R/gpurun_out/r03_bench_traced.json 2> $R/gpurun_out/r03_bench_traced.err
python3 - <<'PY'
import csv, glob, collections, os, json, re
R=os.environ["GRAFT_REPO_ROOT"]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("fetch","write","sq"):
    for f in glob.glob(f"{R}/gpurun_out/r03_pmc_bench_{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name=row["Kernel_Name"]
            name=re.sub(r"\(.*$","",name).replace("void ","").replace("svthip::","").strip()
            if "at::" in name or "rocclr" in name: continue
            key=(name, int(row.get("Grid_Size",0) or 0), int(row.get("Workgroup_Size",0) or 0))
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
out=collections.defaultdict(list)
for (name,grid,wg),c in acc.items():
    e={"grid":grid,"workgroup":wg,"dispatches":max(len(v) for v in c.values())}
    for k,v in c.items(): e[k]=sum(v)/len(v)
    out[name].append(e)
for v in out.values(): v.sort(key=lambda e:(e["grid"],e["workgroup"]))
json.dump(out, open(f"{R}/gpurun_out/r03_pmc_traffic.json","w"), indent=1, sort_keys=True)
for k,v in sorted(out.items()):
    for e in v:
        t=(2*e.get("FETCH_SIZE",0)+e.get("WRITE_SIZE",0))/1e3
        busy=4*e.get("SQ_ACTIVE_INST_VALU",0)/(1024*e["GRBM_GUI_ACTIVE"]/8) if e.get("GRBM_GUI_ACTIVE") else 0
        print(f"{k[:58]:58s} grid {e['grid']:>10d} x{e['dispatches']:<3d} traffic {t:9.1f} MB  valu_busy {busy:5.2f}")
PY
grep -h "svthip" $R/gpurun_out/r03_bench_trace/*/*kernel_stats.csv 2>/dev/null | cut -d, -f1-4 | cut -c1-110 | head -40
cp $(ls $R/gpurun_out/r03_bench_trace/*/*kernel_stats.csv | head -1) $R/gpurun_out/r03_bench_kernel_stats.csv 2>/dev/null
cut -c1-300 $R/gpurun_out/r03_bench_traced.json
