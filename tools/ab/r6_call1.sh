#!/bin/bash
# Round 6, first GPU call: new tests (loss plugins as callables, eval-mode refusals, dry-run bench, warm ragged step), the grid-barrier
# micro-benchmark at the 20x20 layers' workgroup counts (review item 4), a baseline bench line, and consumer-side BatchNorm
# (PLYOLO_LAZY, OPTIN build under _optin/) on YOLOX-x 1280 / YOLOv7 together with the fold limit (review item 3).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_loss_plugins.py tests/test_gpu_submodules.py "tests/test_gpu_network.py::test_bench_two_ranks_share_one_gpu_dry_run" \
  "tests/test_gpu_configs.py::test_ragged_channel_blocks_training_step_on_warm_weights" "tests/test_gpu_configs.py::test_ragged_channel_blocks_end_to_end" \
  -m gpu -x -q -s > $O/r06_tests1.txt 2>&1
tail -5 $O/r06_tests1.txt
( cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip ) > $O/r06_gb_build.txt 2>&1
( timeout 120 tools/micro/grid_barrier 2; timeout 120 tools/micro/grid_barrier 0 ) > $O/r06_grid_barrier.txt 2>&1
tail -3 $O/r06_grid_barrier.txt
timeout 600 python bench.py > $O/r06_bench0.json 2> $O/r06_bench0.err
tail -1 $O/r06_bench0.json | cut -c1-300
OPT=$R/_optin/libplyolo_hip_optin.so
if [ -f $OPT ]; then
  STEPS=10 WARMUP=3 timeout 1500 tools/ab/ab_r5.sh r06_lazy_x 2 "--model yolox_x --size 1280 --batch 16" "-" "PLYOLO_LIB=$OPT" "PLYOLO_LIB=$OPT PLYOLO_LAZY=1" "PLYOLO_LIB=$OPT PLYOLO_LAZY=2" \
     "PLYOLO_LIB=$OPT PLYOLO_LAZY=1 PLYOLO_BNRED_MAX_MB=400" "PLYOLO_BNRED_MAX_MB=400" | tail -12
  STEPS=15 WARMUP=3 timeout 1200 tools/ab/ab_r5.sh r06_lazy_v7 2 "--model yolov7 --size 640 --batch 32" "-" "PLYOLO_LIB=$OPT" "PLYOLO_LIB=$OPT PLYOLO_LAZY=1" "PLYOLO_LIB=$OPT PLYOLO_LAZY=2" | tail -8
fi
