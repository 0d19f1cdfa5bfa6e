# same-box A/B: the round-1 tree (git worktree _r1 at 4e298d0, built separately) against the current tree and its switches
# git worktree add _r1 4e298d0 && make -C _r1/pl_yolo_amd/csrc     (the worktree is not kept in the tree)
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  if [ -d _r1 ]; then (cd _r1 && python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "round-1 build                                        "); fi
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "current default (own main stream, 4 queues, 3 lanes) "
  GPU_MAX_HW_QUEUES=3 PLYOLO_OWN_MAIN=0 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "tuned single-process setup (caller stream, 3 queues)"
  GPU_MAX_HW_QUEUES=4 PLYOLO_OWN_MAIN=0 PLYOLO_HEAD_ONE_LANE=0 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "round-1 stream layout (caller stream, 4 q, 4 lanes) "
  PLYOLO_WG3=0 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "PLYOLO_WG3=0 (phase-alternating 3x3 wgrad)           "
  PLYOLO_PW=0 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "PLYOLO_PW=0 (no pointwise kernel)                    "
  PLYOLO_BN_RED_CAP=1024 PLYOLO_BN_RED_UNR=2 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "BatchNorm reduction 1024 x 2 (round-1 grid)          "
done
