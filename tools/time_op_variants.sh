#!/bin/bash
# Time one op (tools/run_op.py) under libfqhip.so and every variants/libfqhip_*.so (tools/variants.sh), ROUNDS times round-robin.
OP=${1:-kron112}; ROUNDS=${ROUNDS:-2}
cd "$(dirname "$0")/.."
for r in $(seq $ROUNDS); do
  for lib in flatquant_amd/lib/libfqhip.so variants/libfqhip_*.so; do
    printf "%-40s " "$(basename $lib)"; FQHIP_LIB=$PWD/$lib TIME_OP=1 python tools/run_op.py $OP 2>&1 | tail -1
  done
done
