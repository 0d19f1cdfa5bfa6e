for i in 1 2; do
  (cd _r1 && python tools/lane_times.py 2>/dev/null | tail -2 | sed 's/^/r1      /')
  PLYOLO_PW=0 python tools/lane_times.py 2>/dev/null | tail -2 | sed 's/^/PW=0    /'
  PLYOLO_PW=1 python tools/lane_times.py 2>/dev/null | tail -2 | sed 's/^/PW=1    /'
  PLYOLO_PW=1 PLYOLO_PW_KCMAX=128 python tools/lane_times.py 2>/dev/null | tail -2 | sed 's/^/PW KC128/'
done
