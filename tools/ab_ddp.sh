# lane streams shared by all plans (default) vs a private set per plan (PLYOLO_PLAN_STREAMS=1), plain and under a one-rank process group
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
P=29920
ddp() { P=$((P+1)); PLYOLO_BENCH_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 1 --no-cpu-baseline --steps 60 2>/dev/null | run "$1"; }
python -m pytest tests/test_gpu_network.py tests/test_gpu_ddp.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
for i in 1 2 3 4; do
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "plain, shared lane streams               "
  PLYOLO_PLAN_STREAMS=1 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "plain, private streams per plan          "
  ddp "one-rank DDP defaults, shared streams    "
  PLYOLO_PLAN_STREAMS=1 ddp "one-rank DDP defaults, private streams   "
  GPU_MAX_HW_QUEUES=3 PLYOLO_HEAD_ONE_LANE=1 ddp "one-rank DDP 3 queues 3 lanes, shared    "
  GPU_MAX_HW_QUEUES=4 PLYOLO_HEAD_ONE_LANE=1 ddp "one-rank DDP 4 queues 3 lanes, shared    "
done
