#!/bin/bash
# Round 6: the -m gpu suite three more times on one box (flakiness check of the final tree), plus the GPU-idle gap between the plans.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r06_gpu_tests_rep$i.txt 2>&1
  grep -n "passed\|failed" $O/r06_gpu_tests_rep$i.txt | tail -1
done
timeout 300 python tools/glue_times.py > $O/r06_glue_times.txt 2>&1; tail -5 $O/r06_glue_times.txt
