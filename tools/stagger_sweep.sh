# stagger off / automatic on other compiled sizes (the rule was fitted on (13,4), T = 20) -> gpurun_out/stagger.log
set -e
run() { echo -n "QUAD_STAGGER=$1 $2 : " >> gpurun_out/stagger.log
  ALQP_QUAD_STAGGER=$1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/stagger.log; }
for cfg in "--nx 8 --nu 2 --T 10 --batch 16384" "--nx 6 --nu 2 --T 20 --batch 16384" "--nx 2 --nu 1 --T 5 --batch 65536 --variant quad" "--nx 13 --nu 4 --T 50 --batch 16384" "--nx 14 --nu 4 --T 10 --batch 16384"; do
  for i in 1 2; do run 0 "$cfg"; run -1 "$cfg"; done
done
cat gpurun_out/stagger.log
