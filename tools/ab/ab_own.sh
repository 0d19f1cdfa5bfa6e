# PLYOLO_OWN_MAIN=1: lane 0 on a stream of the plan's own, created back to back with the side streams (the default under a distributed launch)
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
P=29990
ddp() { P=$((P+1)); PLYOLO_BENCH_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 1 --no-cpu-baseline --steps 60 2>/dev/null | run "$1"; }
PLYOLO_OWN_MAIN=1 GPU_MAX_HW_QUEUES=4 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tail -3
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "plain default (caller stream = lane 0, q=3)  "
  PLYOLO_OWN_MAIN=1 GPU_MAX_HW_QUEUES=4 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "plain, own main stream, q=4                   "
  ddp "one-rank DDP defaults (own main, q=4, 3 lanes)"
  PLYOLO_OWN_MAIN=0 PLYOLO_HEAD_ONE_LANE=0 ddp "one-rank DDP, caller stream, q=4, lane/level  "
done
