for i in 1 2; do
  for cfg in "PLYOLO_PW=0" "PLYOLO_PW=1" "PLYOLO_PW=1 PLYOLO_PW_KCMAX=128" "PLYOLO_PW=1 PLYOLO_PW_KCMAX=256"; do
    env $cfg python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['value']), round(d['ms_per_step'],3), round(d['nms']['boxes_per_ms']))"
  done
done
