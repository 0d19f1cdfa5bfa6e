#!/bin/bash
# Round 6, fourth GPU call: the whole -m gpu suite of the final tree, smoke(), the default bench line (timed), the one-rank RCCL self-test of the bench.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/r06_gpu_tests_b.txt 2>&1
grep -n "passed\|failed" $O/r06_gpu_tests_b.txt | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.txt 2>&1; tail -3 $O/r06_smoke.txt
/usr/bin/time -v timeout 900 python bench.py > $O/r06_bench_full.json 2> $O/r06_bench_full.err; grep "Elapsed" $O/r06_bench_full.err; tail -1 $O/r06_bench_full.json | cut -c1-200
PLYOLO_BENCH_FORCE_DDP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_bench_ddp1.json 2> $O/r06_bench_ddp1.err
tail -1 $O/r06_bench_ddp1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('per_rank'), d['rccl_world_size'])"
