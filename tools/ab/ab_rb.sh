# same-box A/B of the batched slab fold (PLYOLO_REDUCE_BATCH layers per launch; 1 = one launch per layer)
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3 4 5; do
  for rb in 1 2 4 16; do
    PLYOLO_REDUCE_BATCH=$rb python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "RB=$rb"
  done
done
