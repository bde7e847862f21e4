# team vs quad at batch sizes around the automatic switch (backend.QUAD_MIN_BATCH) -> gpurun_out/crossover.log
set -e
for dt in ${DTYPES:-f32 f64}; do for B in ${BATCHES:-1024 2048 3072 4096 6144}; do for v in team quad; do
  echo -n "$dt B=$B $v " >> gpurun_out/crossover.log
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 --dtype $dt --batch $B --variant $v 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/crossover.log
done; done; done
cat gpurun_out/crossover.log
