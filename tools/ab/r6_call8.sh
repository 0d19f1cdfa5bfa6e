#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
SECONDS=0
timeout 900 python bench.py --profile-out $O/r06_plan_profile_final.json > $O/r06_bench_final.json 2> $O/r06_bench_final.err
echo "default bench.py took ${SECONDS}s"; tail -1 $O/r06_bench_final.json | cut -c1-160
