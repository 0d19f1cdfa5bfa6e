# same-box A/B: the round-1 tree (git worktree _r1 at 4e298d0, built separately) against the current tree and its switches
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
# git worktree add _r1 4e298d0 && make -C _r1/pl_yolo_amd/csrc     (the worktree is not kept in the tree)
for i in 1 2 3; do
  if [ -d _r1 ]; then (cd _r1 && python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "round-1 build                      "); fi
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "current default                    "
  PLYOLO_WG3=0 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "PLYOLO_WG3=0 (phase-alternating 3x3 wgrad)"
  PLYOLO_WG_TRS=3 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "PLYOLO_WG_TRS=3 (tap-row split)    "
  PLYOLO_REDUCE_BATCH=1 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "PLYOLO_REDUCE_BATCH=1              "
  PLYOLO_PW=0 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "PLYOLO_PW=0 (no pointwise kernel)  "
done
