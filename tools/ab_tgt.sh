run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  for t in 256 384 512 768 1024; do
  PLYOLO_WG_TARGET=$t python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "TARGET=$t"
  done
done
