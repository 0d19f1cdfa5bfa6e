# one-rank RCCL group (PLYOLO_BENCH_FORCE_DDP=1): what the data-parallel schedule costs on top of the plain step
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
P=29760
ddp() { P=$((P+1)); PLYOLO_BENCH_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 1 --no-cpu-baseline --steps 60 2>/dev/null | run "$1"; }
for i in 1 2 3 4; do
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "plain (no process group)                      "
  GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "plain, GPU_MAX_HW_QUEUES=8                    "
  ddp "one-rank DDP (exchange started on the wgrad lane, awaited after the plan)"
  PLYOLO_COMM_LANE=5 ddp "one-rank DDP, separate communication lane      "
  GPU_MAX_HW_QUEUES=8 PLYOLO_COMM_LANE=5 ddp "  the same with GPU_MAX_HW_QUEUES=8             "
done
