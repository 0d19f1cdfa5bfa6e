for cfg in "1024 2 32" "2048 2 32" "4096 2 32" "2048 4 32" "2048 1 32" "2048 2 16" "4096 4 16" "4096 2 8"; do
  set -- $cfg
  echo "== cap=$1 unr=$2 div=$3"
  PLYOLO_BN_RED_CAP=$1 PLYOLO_BN_RED_UNR=$2 PLYOLO_BN_RED_DIV=$3 python tools/bench_bn.py 2>&1 | awk '{print $1,$2,$3,$4,$5, "red:",$8,$9}'
done
