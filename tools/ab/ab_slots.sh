# same-box A/B of PLYOLO_STAT_SLOTS (worktrees _s4 / _s16 built with 4 / 16 slots; the tree itself has 8)
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3 4; do
  python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "slots 8 "
  (cd _s4 && python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "slots 4 ")
  (cd _s16 && python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "slots 16")
done
