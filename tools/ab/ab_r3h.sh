# round 3, stream layout robustness: three lanes per plan (main, weight gradients, neck bottom-up + 20x20 level);
# the step from the default stream / a user stream with 0-3 other user streams around, hardware queue counts, a one-rank process group
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for q in 2 3 4 5; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --steps 40 2>/dev/null | run "GPU_MAX_HW_QUEUES=$q, own main stream        "
done
GPU_MAX_HW_QUEUES=3 PLYOLO_OWN_MAIN=0 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | run "q=3, caller stream = lane 0                "
GPU_MAX_HW_QUEUES=4 PLYOLO_OWN_MAIN=0 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | run "q=4, caller stream = lane 0                "
for k in 0 1 2 3 4; do python tools/user_stream_check.py $k 2>/dev/null | tail -1; done
PLYOLO_BENCH_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29977 bench.py --gpus 1 --no-cpu-baseline --steps 40 2>/dev/null | run "one-rank RCCL process group (defaults)      "
python bench.py --no-cpu-baseline --steps 40 2>/dev/null | run "defaults again                             "
