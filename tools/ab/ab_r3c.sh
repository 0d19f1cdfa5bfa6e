# same-box A/B, round 3: lane placements with the weight-gradient lane (1) carrying a head level / the neck in the forward
n=${1:-2}; steps=${2:-60}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "current default (neck 2, head 0,0,2)   "
  PLYOLO_FUSE_BNBWD=0 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PLYOLO_FUSE_BNBWD=0                    "
  PLYOLO_HEAD_LANES=0,1,2 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "head 0,1,2                             "
  PLYOLO_HEAD_LANES=0,2,1 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "head 0,2,1                             "
  PLYOLO_HEAD_LANES=0,1,1 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "head 0,1,1                             "
  PLYOLO_NECK_LANE=1 PLYOLO_HEAD_LANES=0,0,2 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "neck 1, head 0,0,2                     "
  PLYOLO_NECK_LANE=1 PLYOLO_HEAD_LANES=0,2,1 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "neck 1, head 0,2,1                     "
done
