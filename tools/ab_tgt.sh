run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3 4; do
  python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "default"
  PLYOLO_BN_UNR=4 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "unr4"
  PLYOLO_BN_UNR=4 PLYOLO_BN_GRID=512 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "unr4 grid512"
  PLYOLO_BN_UNR=4 PLYOLO_BN_GRID=1024 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "unr4 grid1024"
done
