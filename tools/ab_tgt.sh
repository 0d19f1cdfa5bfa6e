run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "default stream            "
  PLYOLO_BENCH_STREAM=1 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "user-created main stream  "
  PLYOLO_BENCH_STREAM=1 GPU_MAX_HW_QUEUES=4 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "user stream, q=4          "
done
