run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "eager      "
  PLYOLO_GRAPH_FWD=1 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "fwd graph  "
done
python tools/lane_times.py 2>/dev/null | tail -2
PLYOLO_ABLATE_WG=6 python tools/lane_times.py 2>/dev/null | tail -2 | sed 's/^/wgrad-free /'
