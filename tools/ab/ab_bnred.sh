for cfg in "1024 2 32" "512 2 32" "512 4 32" "256 4 32" "256 2 32" "512 4 64" "256 4 128"; do
  set -- $cfg
  echo "== cap=$1 unr=$2 div=$3"
  PLYOLO_BN_RED_CAP=$1 PLYOLO_BN_RED_UNR=$2 PLYOLO_BN_RED_DIV=$3 python tools/bench_bn.py 2>&1 | grep " x " | awk '{print $1,$2,$3,$4,$5, "red:",$8,$9}'
done
