#!/bin/bash
# Round 6, third GPU call: loss-plugin tests, the round's evidence set on the final library, bench lines of the other BASELINE configs.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_loss_plugins.py tests/test_gpu_configs.py::test_ragged_channel_blocks_training_step_on_warm_weights -m gpu -q > $O/r06_tests3.txt 2>&1
tail -3 $O/r06_tests3.txt
timeout 1500 bash tools/collect_evidence.sh r06 > $O/r06_collect.log 2>&1
tail -4 $O/r06_collect.log
timeout 900 python tools/other_configs.py $O/r06_other_configs.md _none > /dev/null 2>&1
grep "^## " $O/r06_other_configs.md
: > $O/r06_plan_walls.txt
for e in "X=0" "PLYOLO_LANES=0" "PLYOLO_DIAG_SKIP_WG=1"; do env $e timeout 300 python tools/plan_walls.py 2>/dev/null | grep "^env" >> $O/r06_plan_walls.txt; done
cat $O/r06_plan_walls.txt
