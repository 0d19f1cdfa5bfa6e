n=${1:-4}; steps=${2:-60}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PLYOLO_REDUCE_BATCH=2 (default)"
  PLYOLO_REDUCE_BATCH=4 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PLYOLO_REDUCE_BATCH=4          "
  PLYOLO_REDUCE_BATCH=3 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PLYOLO_REDUCE_BATCH=3          "
done
