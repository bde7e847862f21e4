set -e
run() { echo "stagger=$1 mode=$2 $3" >> gpurun_out/stagger.log
  ALQP_DEBUG_STAGGER=$1 ALQP_DEBUG_STAGGER_CU=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 $3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/stagger.log; }
run 0 0; run 110 0; run 0 0
cat gpurun_out/stagger.log
