# ceiling experiments: how much of the step would a free weight gradient buy?  (results are wrong under ablation -- timing only)
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2; do
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "default      "
  PLYOLO_ABLATE_WG=6 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "wgrad: no loads/mfma"
  PLYOLO_ABLATE_WG=7 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "wgrad: + no stores  "
  PLYOLO_ABLATE_WG=4 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "wgrad: no mfma      "
  PLYOLO_ABLATE_WG=2 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "wgrad: no loads     "
  PLYOLO_WG_TRS=1 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "TRS=1        "
done
