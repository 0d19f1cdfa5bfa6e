F='head3x3_80|head3x3_40|bb3x3_64'
for a in 50 58 122; do
  echo "== ABL=$a"
  python tools/bench_conv.py "$F" fwd,dgrad "PLYOLO_CONV3WS_ABL=$a" 2>&1 | grep -v "amdgpu.ids\|unpack\|layer\|ENV"
done
