#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python tools/bench_quant.py > $O/r06_quant.txt 2>&1; grep -v amdgpu.ids $O/r06_quant.txt
