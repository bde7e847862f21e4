#!/usr/bin/env python3
"""Condense rocprofv3 csv output of tools/profile.sh into one JSON/text summary."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
summary = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "alqp" in r["Name"]:
            summary.setdefault("kernel_stats", []).append(
                {"name": r["Name"][:60], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                 "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])})
ctr = defaultdict(list)
meta = {}
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_solve_lin" not in r.get("Kernel_Name", ""):
            continue
        ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size", "Accum_VGPR_Count"):
            if k in r:
                meta[k] = r[k]
summary["meta"] = meta
summary["counters_avg_per_dispatch"] = {k: sum(v) / len(v) for k, v in sorted(ctr.items())}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
