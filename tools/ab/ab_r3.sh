# same-box A/B, round 3: the round-2 tree (git worktree _r2 at 0f77289, built separately, not kept in the tree) against the
# current tree and its lane-placement switches.   usage: bash tools/ab/ab_r3.sh [alternations] [steps]
n=${1:-2}; steps=${2:-60}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in $(seq 1 $n); do
  if [ -d _r2 ]; then (cd _r2 && python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "round-2 build                                   "); fi
  python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "current default (neck lane 2, head lanes 0,0,2)  "
  PLYOLO_NECK_LANE=0 PLYOLO_HEAD_LANES=0,2,2 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "round-2 placement (neck 0, head 0,2,2)         "
  PLYOLO_NECK_LANE=2 PLYOLO_HEAD_LANES=0,2,2 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "neck 2, head 0,2,2                             "
  PLYOLO_NECK_LANE=2 PLYOLO_HEAD_LANES=0,0,0 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "neck 2, head 0,0,0                             "
  PLYOLO_NECK_LANE=2 PLYOLO_HEAD_LANES=0,3,2 GPU_MAX_HW_QUEUES=5 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "neck 2, head 0,3,2 (4 lanes)                  "
done
