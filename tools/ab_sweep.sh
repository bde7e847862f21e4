# A/B of two builds of libmi_alqp.so on ONE box (box-to-box spread is 3-5 %): usage
#   bash tools/ab_sweep.sh <lib B> [bench flags]      -> gpurun_out/ab.log
# (the shipped library is A; MI_ALQP_LIB selects B, see deq-mpc-corl_amd/_lib.py)
set -e
B=$1; shift
run() { echo "lib=${1:-shipped} $2" >> gpurun_out/ab.log
  MI_ALQP_LIB=$1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/ab.log; }
for rep in 1 2; do
  run "" "$*"; run "$B" "$*"
done
