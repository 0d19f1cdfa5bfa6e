# EXPERIMENT (needs the PLYOLO_DUMMY_STREAMS hook in api.hip, not kept in the tree): k unused streams created before the lane streams
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2; do
  for k in 0 1 2 3 4; do
    PLYOLO_DUMMY_STREAMS=$k python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "q=3, $k dummy streams before the lanes"
  done
  for k in 0 1 2 3; do
    GPU_MAX_HW_QUEUES=4 PLYOLO_DUMMY_STREAMS=$k python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "q=4, $k dummy streams before the lanes"
  done
done
