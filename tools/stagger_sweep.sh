# stagger off / automatic / explicit units on one box -> gpurun_out/stagger.log   (usage: bash tools/stagger_sweep.sh [bench flags])
set -e
run() { echo "ALQP_QUAD_STAGGER=$1 $2" >> gpurun_out/stagger.log
  ALQP_QUAD_STAGGER=$1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/stagger.log; }
for m in 0 -1 60 140 0 -1; do run $m "$*"; done
