# YOLOX-x 1280x1280 B=16 (MFMA-bound config): which round-3 change costs it 1.5 %?
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],2))"; }
A="--model yolox_x --size 1280 --batch 16 --no-cpu-baseline --steps 15 --warmup 3"
for i in 1 2; do
  (cd _r2 && python bench.py $A 2>/dev/null | run "round-2 build                    ")
  python bench.py $A 2>/dev/null | run "current default                  "
  PLYOLO_FUSE_BNBWD=0 python bench.py $A 2>/dev/null | run "PLYOLO_FUSE_BNBWD=0              "
  PLYOLO_NECK_LANE=0 PLYOLO_HEAD_LANES=0,2,2 python bench.py $A 2>/dev/null | run "round-2 lane placement           "
  PLYOLO_NECK_LANE=0 PLYOLO_HEAD_LANES=0,2,2 PLYOLO_FUSE_BNBWD=0 python bench.py $A 2>/dev/null | run "both off                         "
done
