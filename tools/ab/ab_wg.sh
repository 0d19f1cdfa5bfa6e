run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), [(f['kernel'],round(f['avg_ms']*f['launches_per_step'],3)) for f in d['roofline']['families'][:3]])"; }
for i in 1 2 3; do
  PLYOLO_WG_XCD=0 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "XCD=0"
  PLYOLO_WG_XCD=1 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "XCD=1"
done
