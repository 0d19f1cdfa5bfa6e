# what an initialised NCCL/RCCL process group alone costs per step (no data-parallel schedule), and which knob removes it
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
P=29620
pg() { P=$((P+1)); PLYOLO_BENCH_PG_ONLY=1 PLYOLO_BENCH_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 1 --no-cpu-baseline --steps 60 2>/dev/null | run "$1"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "plain                                  "
  pg "process group only                     "
  TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0 pg "PG, no monitoring / async error handling"
  RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0 pg "PG, MSCCL off                          "
  NCCL_MAX_NCHANNELS=2 NCCL_MIN_NCHANNELS=1 pg "PG, 1-2 channels                       "
done
