#!/bin/bash
# Memory-pipeline counters of the headline launch (texture addresser, L1, L2), one rocprofv3 --pmc
# pass per group (counters only, never mixed with tracing). Usage: tools/profile_mem.sh <outdir> [bench args...]
set -uo pipefail
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
# The TA block has two counter slots per instance: more than two TA_* counters in one pass abort rocprofv3
# ("error code 38: Request exceeds the capabilities of the hardware to collect"), so they go two by two.
FAILED=0
for ctrs in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
            "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
            "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
            "TA_TOTAL_WAVEFRONTS_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr" \
            "TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" \
            "TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_READ_sum TCC_WRITE_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/mem$i" -- python3 bench.py --no-cpu-baseline --no-extras "$@" > "$OUT/mem$i.log" 2>&1 || { echo "mem pass $i failed ($ctrs)"; FAILED=1; }
  echo "pass $i done"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
ctr = defaultdict(list)
for f in glob.glob(os.path.join(out, "mem*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_solve_lin" in r.get("Kernel_Name", ""):
            ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
s = {k: sum(v) / len(v) for k, v in sorted(ctr.items())}
json.dump({"counters_avg_per_dispatch": s}, open(os.path.join(out, "mem_summary.json"), "w"), indent=1)
print(json.dumps(s, indent=1))
PY
exit $FAILED
