run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), [(f['kernel'],round(f['avg_ms']*f['launches_per_step'],3)) for f in d['roofline']['families'][:1]])"; }
export PLYOLO_WG_TRS=3
for i in 1 2; do
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "TRS=3 default      "
  PLYOLO_WG_BUDGET_MB=40 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "TRS=3 budget40     "
  PLYOLO_WG_TARGET=1536 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "TRS=3 target1536   "
  PLYOLO_WG_BUDGET_MB=12 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "TRS=3 budget12     "
done
