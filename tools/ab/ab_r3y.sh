run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3))"; }
A="--model yolox_x --size 1280 --batch 16 --no-cpu-baseline --steps 15 --warmup 3"
for i in 1 2; do
  python bench.py $A 2>/dev/null | run "yolox_x 1280: fused up to 64 MB   "
  PLYOLO_FUSE_BNBWD=0 python bench.py $A 2>/dev/null | run "yolox_x 1280: never fused         "
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "yolox_s: fused up to 64 MB        "
  PLYOLO_FUSE_BNBWD_MB=1000 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "yolox_s: fused everywhere         "
  PLYOLO_FUSE_BNBWD_MB=30 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "yolox_s: fused up to 30 MB        "
  PLYOLO_FUSE_BNBWD=0 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "yolox_s: never fused              "
done
