# one-rank RCCL group (PLYOLO_BENCH_FORCE_DDP=1): what the data-parallel schedule costs on top of the plain step, by hardware-queue
# count and head-lane layout (the defaults under a distributed launch are 4 queues and a lane per head level)
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
P=29880
ddp() { P=$((P+1)); PLYOLO_BENCH_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 1 --no-cpu-baseline --steps 60 2>/dev/null | run "$1"; }
for i in 1 2 3 4; do
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "plain (3 queues, 3 streams)                 "
  ddp "one-rank DDP, defaults (4 queues, lane/level)"
  GPU_MAX_HW_QUEUES=3 PLYOLO_HEAD_ONE_LANE=1 ddp "one-rank DDP, 3 queues, 3 lanes             "
  GPU_MAX_HW_QUEUES=4 PLYOLO_HEAD_ONE_LANE=1 ddp "one-rank DDP, 4 queues, 3 lanes             "
  PLYOLO_COMM_LANE=5 ddp "one-rank DDP, separate communication lane   "
done
