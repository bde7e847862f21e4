# stagger experiments on one box -> gpurun_out/stagger.log. ALQP_QUAD_STAGGER: the shipped control (-1 auto, 0 off, n units);
# ALQP_DEBUG_STAGGER / ALQP_DEBUG_STAGGER_CU: explicit units and grouping mode through AlqpParams.flags (alqp_kernels.hip)
set -e
run() { echo "QUAD_STAGGER=$1 DEBUG_STAGGER=$2 mode=$3 $4" >> gpurun_out/stagger.log
  ALQP_QUAD_STAGGER=$1 ALQP_DEBUG_STAGGER=$2 ALQP_DEBUG_STAGGER_CU=$3 timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 $4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/stagger.log; }
for i in 1 2 3; do run 0 0 0 "$*"; run 0 100 0 "$*"; run 0 100 5 "$*"; run 0 33 5 "$*"; done
cat gpurun_out/stagger.log
