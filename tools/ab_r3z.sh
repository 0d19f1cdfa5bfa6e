# same-box A/B: the tree before the loss-kernel changes of the last part of round 3 (_prev/ = git worktree at 5731154, built
# separately) against the current tree
n=${1:-3}; steps=${2:-60}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in $(seq 1 $n); do
  (cd _prev && python bench.py --no-cpu-baseline --steps $steps 2>/dev/null) | run "before (5731154)"
  python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "current         "
done
