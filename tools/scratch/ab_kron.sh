#!/bin/bash
# A/B of library variants on tools/time_kron.py cases: tools/scratch/ab_kron.sh "<time_kron args>" lib1 lib2 ...
# (variants/libfqhip_<name>.so from tools/variants.sh; "default" = the built library; two rounds, interleaved)
args="$1"; shift
for rep in 1 2; do
  for lib in default "$@"; do
    if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
    echo "== $lib"; python tools/time_kron.py $args
  done
done
