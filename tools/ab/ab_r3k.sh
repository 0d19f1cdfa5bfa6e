n=${1:-3}; steps=${2:-60}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "shortcut gradient forwarded by bn_act_bwd_dz"
  PLYOLO_RES_IN_DZ=0 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PLYOLO_RES_IN_DZ=0 (copy_add launches)     "
done
