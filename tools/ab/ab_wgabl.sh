# standalone ablation of the weight-gradient kernel: which phase holds the time?  (1 no stores, 2 no tile loads after the first, 4 no MFMA phase)
for trs in 1; do
for a in 0 1 2 4 6 7; do
  echo "== TRS=$trs ablate=$a"
  python tools/bench_conv.py 'bb3x3_64|head3x3_80|s2_64_128|s2_32_64|pw_128_128' wgrad "PLYOLO_WG_TRS=$trs,PLYOLO_ABLATE_WG=$a" 2>&1 | grep -v "amdgpu.ids\|unpack\|layer\|ENV" | awk '{print $1, $(NF-2)}'
done
done
for S in 64 128 192 256; do echo "== TRS=1 S=$S";  python tools/bench_conv.py 'bb3x3_64|head3x3_80' wgrad "PLYOLO_WG_TRS=1,PLYOLO_WG_S=$S" 2>&1 | grep -v "amdgpu.ids\|unpack\|layer\|ENV" | awk '{print $1, $(NF-2)}'; done
