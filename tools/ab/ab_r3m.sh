# v_mfma_f32_16x16x32_bf16 in the 3x3 forward / data-gradient kernel: same-box A/B on the whole step
n=${1:-3}; steps=${2:-60}
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "32x32x16 (default)        "
  PLYOLO_MFMA16=1 python bench.py --no-cpu-baseline --steps $steps 2>/dev/null | run "16x16x32 (PLYOLO_MFMA16=1)"
done
