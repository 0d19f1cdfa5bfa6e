#!/bin/bash
# Same-box A/B of environment variants on another config: tools/ab/ab_cfg.sh <tag> <reps> "<bench args>" "ENV.." "ENV.." ...  ("-" = default)
tag=$1; reps=$2; args=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
out=$O/${tag}_ab.txt; : > $out
for r in $(seq 1 $reps); do
  i=0
  for v in "$@"; do
    i=$((i+1)); e=""; [ "$v" != "-" ] && e="$v"
    ms=$(env $e python bench.py $args --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print('%.3f' % json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
    echo "[$args] rep $r variant $i [$v] ms_per_step $ms" | tee -a $out
  done
done
