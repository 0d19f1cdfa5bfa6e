#!/bin/bash
# Same-box A/B of environment variants on any bench configuration (round 5):
#   tools/ab/ab_r5.sh <tag> <reps> "<bench args>" "ENV1=.. ENV2=.." "ENV3=.." ...     ("-" = the default environment)
# Per variant: <reps> alternating bench lines (ms/step); PROFILE=1 adds one per-family plan profile per variant.
tag=$1; reps=$2; bargs=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
out=$O/${tag}_ab.txt; : > $out
echo "# bench args: $bargs" >> $out
for r in $(seq 1 $reps); do
  i=0
  for v in "$@"; do
    i=$((i+1)); e=""; [ "$v" != "-" ] && e="$v"
    ms=$(env $e python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline $bargs 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
    echo "rep $r variant $i [$v] ms_per_step $ms" | tee -a $out
  done
done
if [ -n "$PROFILE" ]; then
  i=0
  for v in "$@"; do
    i=$((i+1)); e=""; [ "$v" != "-" ] && e="$v"
    env $e python bench.py --steps 6 --warmup 3 --no-cpu-baseline $bargs --profile-out $O/${tag}_prof$i.json > /dev/null 2>&1
    echo "variant $i [$v] profile:" >> $out
    python tools/prof_rows.py $O/${tag}_prof$i.json | head -${PROF_ROWS:-24} >> $out
  done
fi
