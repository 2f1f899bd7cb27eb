for r in 1 2; do
  echo default; timeout 200 python tools/bench_shapes.py 2>&1 | grep -E "int4 linear"
  echo "$1"; FQHIP_LIB=$1 timeout 200 python tools/bench_shapes.py 2>&1 | grep -E "int4 linear"
done
