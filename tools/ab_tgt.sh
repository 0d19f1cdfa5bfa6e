run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
python -m pytest tests/test_gpu_network.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3 4 5; do
  PLYOLO_PACK_SPLIT=0 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "pack up front"
  PLYOLO_PACK_SPLIT=4 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "pack split 4 "
  PLYOLO_PACK_SPLIT=2 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "pack split 2 "
done
python tools/lane_times.py 2>/dev/null | tail -2
