#!/bin/bash
# Same-box A/B of the current tree against the round-3 head (_r3/: git worktree at 2f83f23, built separately): YOLOX-s default bench,
# alternating, then the other BASELINE configs.   tools/ab/ab_r4g.sh <reps>   (run through gpurun)
reps=${1:-3}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
out=$O/r04_ab_vs_r3.txt; : > $out
ms() { (cd $1 && python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f ms/step %.0f img/s' % (d['ms_per_step'], d['value']))"); }
for r in $(seq 1 $reps); do
  echo "rep $r round-3 head: $(ms $R/_r3)" | tee -a $out
  echo "rep $r current:      $(ms $R)" | tee -a $out
done
python tools/other_configs.py $O/r04_other_configs.md _r3 > /dev/null 2>&1
grep "^## " $O/r04_other_configs.md | tee -a $out
