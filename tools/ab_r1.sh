# same-box A/B: the round-1 tree (git worktree _r1 at 4e298d0, built separately) against the current tree and its switches
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  (cd _r1 && python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "round-1 build                      ")
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "current default                    "
  PLYOLO_WG3=0 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "PLYOLO_WG3=0 (phase-alternating 3x3 wgrad)"
  PLYOLO_WG_TRS=3 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "PLYOLO_WG_TRS=3 (tap-row split)    "
  PLYOLO_REDUCE_BATCH=1 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "PLYOLO_REDUCE_BATCH=1              "
  PLYOLO_PW=0 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "PLYOLO_PW=0 (no pointwise kernel)  "
done
