#!/usr/bin/env python3
"""Lists the kernels of one translation unit that spill vector registers, and flags the pattern that corrupted a test in
round 3: a spill store immediately in front of the `s_or_b64 exec` that opens a divergent region's exit block (the store
then runs under the region's EXEC mask - possibly 0 - and loses lanes).
    python tools/spill_report.py [part 1|2|3] [extra hipcc flags...]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "deq-mpc-corl_amd", "csrc")
part = sys.argv[1] if len(sys.argv) > 1 else "1"
out = os.path.join(tempfile.gettempdir(), f"alqp_part{part}.s")
cmd = ["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-mllvm",
       "-pragma-unroll-threshold=1000000", f"-DALQP_PART={part}", "-S", "--cuda-device-only",
       os.path.join(CSRC, "alqp_kernels.hip"), "-o", out] + sys.argv[2:]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
lines = txt.split("\n")
cur, hits = None, {}
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1)
    if "Folded Spill" in l and "scratch_store" in l:
        nxt = next((x for x in lines[i + 1:i + 4] if x.strip() and not x.strip().startswith(";")), "")
        if re.search(r"s_or_b64\s+exec", nxt):
            hits.setdefault(cur, []).append(i + 1)
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.vgpr_spill_count:\s+(\d+)", txt, re.S):
    name, body, sp = m.group(1), m.group(2), int(m.group(3))
    if sp:
        vg = re.search(r"\.vgpr_count:\s+(\d+)", body).group(1)
        print(f"{sp:4d} spills  {vg} VGPRs  {name[:100]}  {'SPILL IN FRONT OF EXEC RESTORE at lines ' + str(hits[name]) if name in hits else ''}")
print("kernels with the pattern:", len(hits))
