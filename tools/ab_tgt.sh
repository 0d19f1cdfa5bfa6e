run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3 4; do
  python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "default     "
  PLYOLO_WGRAD_BATCH=2 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "WGRAD_BATCH=2"
  PLYOLO_WGRAD_BATCH=4 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "WGRAD_BATCH=4"
  PLYOLO_PW_KCMAX=128 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "PW KC128    "
done
