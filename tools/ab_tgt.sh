run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3 4; do
  python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "q=3, head levels on lanes 0/2/3"
  PLYOLO_HEAD_ONE_LANE=1 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "q=3, head levels on lanes 0/2/2"
  PLYOLO_HEAD_ONE_LANE=1 GPU_MAX_HW_QUEUES=4 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "q=4, head levels on lanes 0/2/2"
  PLYOLO_HEAD_ONE_LANE=1 GPU_MAX_HW_QUEUES=2 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "q=2, head levels on lanes 0/2/2"
done
