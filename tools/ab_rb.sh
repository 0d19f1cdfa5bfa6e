# same-box A/B of the batched slab fold (PLYOLO_REDUCE_BATCH layers per launch; 1 = one launch per layer)
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
python -m pytest tests/test_gpu_network.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
  for rb in 1 4 8 16 999; do
    PLYOLO_REDUCE_BATCH=$rb python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "RB=$rb"
  done
done
