cd $GRAFT_REPO_ROOT
STEPS=10 WARMUP=3 bash tools/ab_r5.sh r05j_x 2 "--model yolox_x --size 1280 --batch 16" - PLYOLO_BNRED_MAX_MB=160 PLYOLO_BNRED_MAX_MB=400 PLYOLO_BNRED_MAX_MB=100000 > /dev/null
STEPS=15 WARMUP=3 bash tools/ab_r5.sh r05j_v7 2 "--model yolov7 --batch 32" - PLYOLO_BNRED_MAX_MB=160 PLYOLO_BNRED_MAX_MB=400 PLYOLO_BNRED_MAX_MB=100000 > /dev/null
STEPS=15 WARMUP=3 bash tools/ab_r5.sh r05j_l 2 "--model yolox_l --batch 16" - PLYOLO_BNRED_MAX_MB=160 PLYOLO_BNRED_MAX_MB=100000 > /dev/null
cat gpurun_out/r05j_*_ab.txt
