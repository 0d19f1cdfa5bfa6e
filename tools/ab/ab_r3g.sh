# pruned library (no opt-in instances) vs the round-2 build and the previous commit's library
n=${1:-2}; steps=${2:-60}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in $(seq 1 $n); do
  if [ -d _r2 ]; then (cd _r2 && python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "round-2 build                         "); fi
  python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "current (library without opt-in paths)"
done
