run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2; do
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "default              "
  PLYOLO_WG_TARGET=512 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "WG_TARGET=512        "
  PLYOLO_WG_TARGET=384 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "WG_TARGET=384        "
  PLYOLO_WG_BUDGET_MB=12 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "WG_BUDGET_MB=12      "
  PLYOLO_WG_BUDGET_MB=24 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "WG_BUDGET_MB=24      "
  PLYOLO_WG_BUDGET1_MB=8 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "WG_BUDGET1_MB=8 (1x1)"
  PLYOLO_REDUCE_BATCH=6 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "REDUCE_BATCH=6       "
  PLYOLO_BN_RED_CAP=384 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "BN_RED_CAP=384       "
  PLYOLO_BN_GRID=768 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "BN_GRID=768          "
done
