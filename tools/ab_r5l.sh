cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_elementwise.py -q -x 2>&1 | tail -2
L=$GRAFT_REPO_ROOT/_optin/libplyolo_nopf.so
bash tools/ab_r5.sh r05l 3 "" PLYOLO_LIB=$L - > /dev/null
cat gpurun_out/r05l_ab.txt
