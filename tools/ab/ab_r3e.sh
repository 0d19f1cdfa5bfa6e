n=${1:-2}; steps=${2:-60}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "BN streams: 2 rows requested before the prologue (<= 64 MB) "
  PLYOLO_BN_PF=0 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PLYOLO_BN_PF=0                                              "
  PLYOLO_BN_PF_MB=30 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PLYOLO_BN_PF_MB=30                                          "
  PLYOLO_BN_PF_MB=1000 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PLYOLO_BN_PF_MB=1000                                        "
done
