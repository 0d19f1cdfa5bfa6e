run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "default                 "
  PLYOLO_WG_BUDGET_MB=12 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "slab budget 12 MB       "
  PLYOLO_WG_BUDGET_MB=20 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "slab budget 20 MB       "
  PLYOLO_BN_RED_CAP=256 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "BN reduce 256 WGs       "
  PLYOLO_BN_GRID=512 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "BN fwd/dz grid 512      "
  PLYOLO_PW_KCMAX=128 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "pointwise KC128         "
  PLYOLO_WG_TRS=3 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "wgrad tap-row split     "
done
