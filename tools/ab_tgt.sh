run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=[x for x in d['roofline']['families'] if x['kernel']=='conv_wgrad_kernel']; print('$1', round(d['value']), round(d['ms_per_step'],3), round(f[0]['avg_ms']*f[0]['launches_per_step'],3) if f else None)"; }
for i in 1 2 3 4; do
  python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "1x1 budget 16"
  PLYOLO_WG_BUDGET1_MB=8 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "1x1 budget 8 "
  PLYOLO_WG_BUDGET1_MB=32 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "1x1 budget 32"
  PLYOLO_WG_BUDGET1_MB=64 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | run "1x1 budget 64"
done
