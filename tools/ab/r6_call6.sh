#!/bin/bash
# Round 6 experiment: a lighter slab fold (fewer workgroups per reduce_slabs launch, same sums) -- library built from the tree + one knob, _optin/libplyolo_fold.so
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$R/_optin/libplyolo_fold.so
STEPS=30 WARMUP=5 timeout 1500 tools/ab/ab_r5.sh r06_fold 3 "" "-" "PLYOLO_LIB=$L" "PLYOLO_LIB=$L PLYOLO_FOLD_COLS=128" "PLYOLO_LIB=$L PLYOLO_FOLD_COLS=64" "PLYOLO_LIB=$L PLYOLO_FOLD_COLS=32" "PLYOLO_LIB=$L PLYOLO_FOLD_COLS=16"
