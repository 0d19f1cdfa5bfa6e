run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), [(f['kernel'],round(f['avg_ms']*f['launches_per_step'],3)) for f in d['roofline'].get('families',[])[:1]])"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "TRS=1 B0=20"
  PLYOLO_WG_BUDGET0_MB=28 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "TRS=1 B0=28"
  PLYOLO_WG_BUDGET0_MB=38 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "TRS=1 B0=38"
  PLYOLO_WG_BUDGET0_MB=14 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "TRS=1 B0=14"
  PLYOLO_REDUCE_BATCH=1 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "TRS=1 RB=1"
  (cd _r1 && python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "round 1 ")
done
