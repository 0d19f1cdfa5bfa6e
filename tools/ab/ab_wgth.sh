python -m pytest tests/test_gpu_conv.py tests/test_gpu_network.py -x -q -m gpu 2>&1 | tail -3
F='head3x3|bb3x3|stem|s2_|pw_'
for e in "X=0" "PLYOLO_WG_TRS=1" "PLYOLO_WG_TRS=1,PLYOLO_WG_BUDGET_MB=40"; do
  python tools/bench_conv.py "$F" wgrad "$e" 2>&1 | grep -v "amdgpu.ids\|unpack" | awk '{print $1, $(NF-2), $(NF-1), $NF}'
done
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), [(f['kernel'],round(f['avg_ms']*f['launches_per_step'],3)) for f in d['roofline'].get('families',[])[:1]])"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "default "
  PLYOLO_WG_TRS=1 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "TRS=1"
  PLYOLO_WG_TRS=1 PLYOLO_WG_BUDGET_MB=40 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "TRS=1 B40"
  (cd _r1 && python bench.py --no-cpu-baseline --steps 60 2>/dev/null | run "round 1 ")
done
