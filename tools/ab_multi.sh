# A/B/C... of several builds of libmi_alqp.so on ONE box (box-to-box spread is 3-5 %), two interleaved repetitions:
#   bash tools/ab_multi.sh <out.log> <lib1> <lib2> ... -- [bench flags]       ("" or "shipped" = the shipped library)
set -e
OUT=$1; shift
LIBS=()
while [ "$#" -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$#" -gt 0 ] && shift
run() { echo -n "lib=${1:-shipped} : " >> "$OUT"
  L=$1; [ "$L" = "shipped" ] && L=""
  MI_ALQP_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> "$OUT"; }
for rep in 1 2; do
  for l in "${LIBS[@]}"; do run "$l" "$@"; done
done
cat "$OUT"
