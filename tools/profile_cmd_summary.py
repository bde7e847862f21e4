#!/usr/bin/env python3
"""Condense the rocprofv3 csv output of tools/profile_cmd.sh into <outdir>/summary.json: per kernel whose name
contains the given substring, time statistics and counter averages per dispatch."""
import csv, glob, json, os, sys
from collections import defaultdict

out, ksub = sys.argv[1], sys.argv[2]
summary = {"kernel_filter": ksub, "kernels": {}}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r["Name"]:
            summary["kernels"].setdefault(r["Name"][:100], {})["time"] = {
                "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
ctr = defaultdict(lambda: defaultdict(list))
meta = defaultdict(dict)
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if ksub not in name:
            continue
        name = name[:100]
        ctr[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size", "Accum_VGPR_Count"):
            if k in r:
                meta[name][k] = r[k]
for name in ctr:
    d = summary["kernels"].setdefault(name, {})
    d["meta"] = meta[name]
    d["counters_avg_per_dispatch"] = {k: sum(v) / len(v) for k, v in sorted(ctr[name].items())}
    d["dispatches_seen"] = max(len(v) for v in ctr[name].values())
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
