#!/bin/bash
# Round 6, second GPU call: the new tests again (name clash fixed), the two-level grid barrier, then the whole -m gpu suite.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_loss_plugins.py tests/test_gpu_submodules.py "tests/test_gpu_network.py::test_bench_two_ranks_share_one_gpu_dry_run" \
  "tests/test_gpu_configs.py::test_ragged_channel_blocks_training_step_on_warm_weights" "tests/test_gpu_configs.py::test_ragged_channel_blocks_end_to_end" \
  -m gpu -q -s > $O/r06_tests2.txt 2>&1
tail -15 $O/r06_tests2.txt
( cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip ) > $O/r06_gb_build.txt 2>&1
( timeout 120 tools/micro/grid_barrier 2 ) > $O/r06_grid_barrier2.txt 2>&1
grep "two-level" $O/r06_grid_barrier2.txt | grep "K 8"
timeout 1500 python -m pytest tests -m gpu -q -x > $O/r06_gpu_tests_a.txt 2>&1
tail -5 $O/r06_gpu_tests_a.txt
