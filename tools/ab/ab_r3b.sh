# same-box A/B, round 3: fused BatchNorm-backward in the pointwise data gradient, selective lazy activations
n=${1:-2}; steps=${2:-60}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in $(seq 1 $n); do
  if [ -d _r2 ]; then (cd _r2 && python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "round-2 build                          "); fi
  python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "current default                        "
  PLYOLO_FUSE_BNBWD=0 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PLYOLO_FUSE_BNBWD=0                    "
  PLYOLO_FUSE_BNBWD_BLK=1 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PLYOLO_FUSE_BNBWD_BLK=1                "
  PLYOLO_FUSE_BNBWD_BLK=4 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PLYOLO_FUSE_BNBWD_BLK=4                "
  PLYOLO_LAZY=2 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PLYOLO_LAZY=2 (pointwise consumers)    "
done
