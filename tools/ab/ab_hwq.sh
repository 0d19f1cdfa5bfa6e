# GPU_MAX_HW_QUEUES (ROCclr: hardware queues the process's streams are mapped onto; runtime default 4, this package's default 3)
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "unset (bench.py sets 3)"
  for q in 1 2 3 4 5 6; do
    GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "GPU_MAX_HW_QUEUES=$q    "
  done
done
