# ROCm runtime knobs (environment, read when the runtime starts): do any of them move the step?
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "default                    "
  HIP_FORCE_DEV_KERNARG=1 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "HIP_FORCE_DEV_KERNARG=1    "
  HIP_FORCE_DEV_KERNARG=0 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "HIP_FORCE_DEV_KERNARG=0    "
  HSA_ENABLE_INTERRUPT=0 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "HSA_ENABLE_INTERRUPT=0     "
  GPU_STREAMOPS_CP_WAIT=1 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "GPU_STREAMOPS_CP_WAIT=1    "
  ROC_EVENT_NO_FLUSH=1 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "ROC_EVENT_NO_FLUSH=1       "
  HSA_ENABLE_SDMA=0 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | run "HSA_ENABLE_SDMA=0          "
done
