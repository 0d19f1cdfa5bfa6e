# persistent weight-resident small-channel 3x3 tiles: per-launch and in-step A/B (one box)
python tools/bench_conv.py "stem|s2_32_64|bb3x3_32_160" fwd,dgrad "PLYOLO_PERS=0;PLYOLO_PERS=1,PLYOLO_PERS_WG=1;PLYOLO_PERS=1,PLYOLO_PERS_WG=2;PLYOLO_PERS=1,PLYOLO_PERS_WG=4"
python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -2
n=${1:-3}; steps=${2:-60}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in $(seq 1 $n); do
  PLYOLO_PERS=0 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PERS=0     "
  python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PERS=1 WG=2"
  PLYOLO_PERS_WG=1 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "PERS=1 WG=1"
done
