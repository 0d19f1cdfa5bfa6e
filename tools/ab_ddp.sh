# one-rank RCCL group (PLYOLO_BENCH_FORCE_DDP=1): what the data-parallel schedule costs on top of the plain step
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
P=29580
ddp() { P=$((P+1)); PLYOLO_BENCH_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 1 --no-cpu-baseline --steps 60 2>/dev/null | run "$1"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "plain                                   "
  ddp "one-rank DDP                            "
  PLYOLO_DDP_DBG=1 ddp "DBG=1: no communication lane            "
done
