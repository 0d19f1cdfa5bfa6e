n=${1:-2}; steps=${2:-60}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "default                                    "
  PLYOLO_HEAD_LANES_FWD=0,1,2 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "forward plan: head level 1 on lane 1       "
  PLYOLO_HEAD_LANES_FWD=0,2,1 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "forward plan: head levels 0,2,1            "
  PLYOLO_HEAD_LANES_FWD=0,1,1 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "forward plan: head levels 0,1,1            "
done
python tools/lane_times.py 2>/dev/null | tail -2
PLYOLO_HEAD_LANES_FWD=0,1,2 python tools/lane_times.py 2>/dev/null | tail -2
