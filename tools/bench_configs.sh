#!/bin/bash
# Throughput of the BASELINE.json configs on one MI355X (both variants, f32/f64). Output: gpurun_out/configs.jsonl
set -uo pipefail
OUT=${1:-gpurun_out/configs.jsonl}
: > "$OUT"
run() { python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" >> "$OUT" 2>/dev/null || echo "{\"failed\": \"$*\"}" >> "$OUT"; }
for v in quad team; do
  for d in f32 f64; do
    run --variant $v --dtype $d --batch 4096 --T 5 --nx 2 --nu 1
    run --variant $v --dtype $d --batch 8192 --T 10 --nx 8 --nu 2
    run --variant $v --dtype $d --batch 16384 --T 20 --nx 13 --nu 4
    run --variant $v --dtype $d --batch 8192 --T 50 --nx 13 --nu 4
    run --variant $v --dtype $d --batch 16384 --T 20 --nx 12 --nu 4
  done
done
python3 - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    if "failed" in d: print("FAILED", d["failed"]); continue
    print(f'{d["config"]["kernel_variant"]:5s} {d["dtype"]} {d["config"]["workload"][:52]:52s} {d["value"]/1e6:8.3f} M solves/s  {d["ms_per_step"]:8.3f} ms  ok={d["all_instances_ok"]} hbm_frac={d["roofline"]["frac"]:.4f}')
PY
