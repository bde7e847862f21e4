#!/bin/bash
# rocprofv3 over the FULL bench line (headline + f64 + newton_step + ipm sub-records): one kernel-trace pass, then
# FETCH_SIZE / WRITE_SIZE passes (counters only). Usage (GPU box): tools/profile_all.sh <outdir>
set -uo pipefail
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/trace.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/pmc_$c.log" 2>&1 || echo "pmc pass $c failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
stats = {}
for f in glob.glob(out + "/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "alqp" in r["Name"]:
            stats[r["Name"][:70]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + f"/pmc_{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "alqp" in r["Kernel_Name"] and r["Counter_Name"] == c:
                pmc[r["Kernel_Name"][:70]][c].append(float(r["Counter_Value"]))
for k, v in stats.items():
    if k in pmc and pmc[k]["FETCH_SIZE"] and pmc[k]["WRITE_SIZE"]:
        fe = sum(pmc[k]["FETCH_SIZE"]) / len(pmc[k]["FETCH_SIZE"]); wr = sum(pmc[k]["WRITE_SIZE"]) / len(pmc[k]["WRITE_SIZE"])
        v.update(FETCH_SIZE_KB=fe, WRITE_SIZE_KB=wr, hbm_GB_per_launch=(2 * fe + wr) * 1024 / 1e9,
                 TBps=(2 * fe + wr) * 1024 / 1e12 / (v["avg_ms"] * 1e-3))
json.dump(stats, open(out + "/kernels_summary.json", "w"), indent=1)
print(json.dumps(stats, indent=1))
PY
