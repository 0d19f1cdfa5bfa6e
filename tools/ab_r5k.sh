cd $GRAFT_REPO_ROOT
python bench.py --steps 6 --warmup 3 --no-cpu-baseline --model yolov7 --batch 32 --profile-out gpurun_out/r05k_v7_prof.json > /dev/null 2>&1
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --model yolox_x --size 1280 --batch 16 --profile-out gpurun_out/r05k_x_prof.json > /dev/null 2>&1
python tools/bench_wgrad1.py > gpurun_out/r05k_wgrad1.txt 2>&1
ls -la gpurun_out/
