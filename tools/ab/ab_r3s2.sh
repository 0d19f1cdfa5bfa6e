steps=${1:-60}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
arm() { env "$@" python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "$(printf '%-44s' "$*")"; }
for i in 1 2 3; do
  arm X=0
  arm PLYOLO_REDUCE_BATCH=6
  arm PLYOLO_REDUCE_BATCH=8
  arm PLYOLO_WGRAD_BATCH=2
  arm PLYOLO_WGRAD_BATCH=4
  arm PLYOLO_BN_UNR=4
  arm PLYOLO_BN_RED_DIV=64
  arm PLYOLO_BN_RED_DIV=16
done
