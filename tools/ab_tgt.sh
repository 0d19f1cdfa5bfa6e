run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "3 streams, q=3 (default)     "
  PLYOLO_HEAD_ONE_LANE=2 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "2 streams, q=3               "
  PLYOLO_HEAD_ONE_LANE=2 GPU_MAX_HW_QUEUES=2 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "2 streams, q=2               "
  PLYOLO_HEAD_ONE_LANE=0 GPU_MAX_HW_QUEUES=4 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "4 streams, q=4 (old default) "
done
