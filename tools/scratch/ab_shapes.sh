# A/B of a variant library on the non-64x64 Kronecker shapes: default vs FQHIP_LIB=$1 (two rounds each, interleaved)
for r in 1 2; do
  echo "default"; KRON_ONLY=1 timeout 100 python tools/bench_shapes.py 2>&1 | grep -E "d=8192|d=7168|d=2048"
  echo "$1"; FQHIP_LIB=$1 KRON_ONLY=1 timeout 100 python tools/bench_shapes.py 2>&1 | grep -E "d=8192|d=7168|d=2048"
done
