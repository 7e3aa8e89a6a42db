#!/usr/bin/env python3
"""Post-processing of tools/run_r03_pmc_traffic.sh: the counter_collection CSVs of the separate rocprofv3 --pmc passes ->
<out>.json  {kernel: [ {grid, workgroup, dispatches, FETCH_SIZE, WRITE_SIZE (KB per launch), GRBM_GUI_ACTIVE, SQ_* ...} by grid ]}.
usage: pmc_traffic_summary.py <gpurun_out dir> <prefix of the pass directories> <out.json>"""
import collections
import csv
import glob
import json
import re
import sys

root, prefix, out_path = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/{prefix}*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*$", "", name).replace("void ", "").replace("svthip::", "").strip()
        if not name or "at::" in name or "rocclr" in name:
            continue
        key = (name, int(row.get("Grid_Size", 0) or 0), int(row.get("Workgroup_Size", 0) or 0))
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = collections.defaultdict(list)
for (name, grid, wg), c in acc.items():
    e = {"grid": grid, "workgroup": wg, "dispatches": max(len(v) for v in c.values())}
    for k, v in c.items():
        e[k] = sum(v) / len(v)
    out[name].append(e)
for v in out.values():
    v.sort(key=lambda e: (e["grid"], e["workgroup"]))
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items()):
    for e in v:
        t = (2 * e.get("FETCH_SIZE", 0) + e.get("WRITE_SIZE", 0)) / 1e3
        busy = 4 * e.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * e["GRBM_GUI_ACTIVE"] / 8) if e.get("GRBM_GUI_ACTIVE") else 0
        print(f"{k[:58]:58s} grid {e['grid']:>10d} wg {e['workgroup']:>4d} x{e['dispatches']:<3d} traffic {t:9.1f} MB  valu_busy {busy:5.2f}")
