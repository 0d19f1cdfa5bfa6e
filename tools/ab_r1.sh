# same-box A/B: the round-1 tree (git worktree _r1 at 4e298d0, built separately) against variants of the current tree
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2; do
  (cd _r1 && python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "r1              ")
  PLYOLO_PW=0 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "now PW=0        "
  PLYOLO_PW_LOOP=0 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "plain KC64      "
  PLYOLO_PW_LOOP=0 PLYOLO_PW_KCMAX=128 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "plain KC128     "
  PLYOLO_PW_LOOP=0 PLYOLO_PW_KCMAX=256 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "plain KC256     "
  PLYOLO_PW_LOOP=1 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "pipe  KC64      "
  PLYOLO_PW_LOOP=1 PLYOLO_PW_KCMAX=128 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "pipe  KC128     "
done
