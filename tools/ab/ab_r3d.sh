n=${1:-2}; steps=${2:-60}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "fused BN-bwd dgrad, 64-row tiles, KC 64 "
  PLYOLO_BNB_KC=32 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "fused, KC 32                            "
  PLYOLO_FUSE_BNBWD=0 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PLYOLO_FUSE_BNBWD=0                     "
  PLYOLO_FUSE_BNBWD_BLK=4 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "fused, up to 4 blocks                   "
done
