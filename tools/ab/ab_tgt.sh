run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "default: own main, 4 queues, 3 lanes"
  PLYOLO_HEAD_ONE_LANE=0 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "own main, 4 queues, 4 lanes         "
  PLYOLO_HEAD_ONE_LANE=2 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "own main, 4 queues, 2 lanes         "
done
