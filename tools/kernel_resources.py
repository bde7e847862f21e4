#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr log)."""
import re
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
K = {"sgpr": r"SGPRs", "vgpr": r"VGPRs", "agpr": r"AGPRs",
     "scratch": r"ScratchSize \[bytes/lane\]", "occ": r"Occupancy \[waves/SIMD\]",
     "lds": r"LDS Size \[bytes/block\]"}
for b in blocks:
    name = b.split("\n")[0].strip()
    vals = {}
    for k, pat in K.items():
        m = re.search(pat + r": (\d+)", b)
        vals[k] = m.group(1) if m else "?"
    print(f"{name[:70]:72s} " + " ".join(f"{k}={v}" for k, v in vals.items()))
