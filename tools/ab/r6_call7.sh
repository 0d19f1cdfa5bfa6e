#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python tools/bench_overlap.py > $O/r06_overlap.txt 2>&1; cat $O/r06_overlap.txt | tail -10
