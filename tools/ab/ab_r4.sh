#!/bin/bash
# Same-box A/B of environment variants (round 4): tools/ab/ab_r4.sh <tag> <reps> "ENV1=.. ENV2=.." "ENV3=.." ...
# ("-" = the default environment).  Per variant: <reps> alternating bench lines (ms/step), then one plan profile + lane times.
tag=$1; reps=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
out=$O/${tag}_ab.txt; : > $out
for r in $(seq 1 $reps); do
  i=0
  for v in "$@"; do
    i=$((i+1)); e=""; [ "$v" != "-" ] && e="$v"
    ms=$(env $e python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
    echo "rep $r variant $i [$v] ms_per_step $ms" | tee -a $out
  done
done
i=0
for v in "$@"; do
  i=$((i+1)); e=""; [ "$v" != "-" ] && e="$v"
  env $e python bench.py --steps 10 --warmup 5 --no-cpu-baseline --profile-out $O/${tag}_prof$i.json > /dev/null 2>&1
  echo "variant $i [$v] profile:" >> $out
  python tools/prof_rows.py $O/${tag}_prof$i.json >> $out
  env $e python tools/lane_times.py 2>/dev/null | tail -2 >> $out
done
tail -60 $out
