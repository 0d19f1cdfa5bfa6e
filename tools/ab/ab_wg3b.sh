run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), [(f['kernel'],round(f['avg_ms']*f['launches_per_step'],3)) for f in d['roofline'].get('families',[])[:1]])"; }
for i in 1 2 3 4; do
  for b in 4 6 8 10 12 16 20; do
  PLYOLO_WG_BUDGET_MB=$b python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "B$b"
  done
done
