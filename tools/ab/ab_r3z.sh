# same-box A/B: the tree before the loss-kernel changes of the last part of round 3 (_prev/ = git worktree at the commit named in $3, built
# separately) against the current tree
n=${1:-3}; steps=${2:-60}; lab=${3:-5731154}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in $(seq 1 $n); do
  (cd _prev && python bench.py --no-cpu-baseline --steps $steps 2>/dev/null) | run "before ($lab)      "
  python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "current         "
done
