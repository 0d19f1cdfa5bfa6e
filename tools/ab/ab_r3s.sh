# round 3: re-tune the round-2 knobs under the new lane placement (one box, every arm twice, interleaved)
steps=${1:-60}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
arm() { env "$@" python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "$(printf '%-44s' "$*")"; }
for i in 1 2; do
  arm X=0
  arm PLYOLO_FORCE_TH=8
  arm PLYOLO_DB=2
  arm PLYOLO_BN_RED_CAP=256
  arm PLYOLO_BN_RED_CAP=1024 PLYOLO_BN_RED_UNR=2
  arm PLYOLO_BN_GRID=512
  arm PLYOLO_BN_GRID=2048
  arm PLYOLO_WG_BUDGET_MB=12
  arm PLYOLO_WG_BUDGET_MB=24
  arm PLYOLO_REDUCE_BATCH=1
  arm PLYOLO_REDUCE_BATCH=4
  arm PLYOLO_PW_KCMAX=128
  arm PLYOLO_CK_MODE=1
  arm PLYOLO_WG_TARGET=1024
  arm PLYOLO_WG_TARGET=512
done
