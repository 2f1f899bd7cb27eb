#!/bin/bash
# A/B of library variants on the Linear4bit GEMM table: tools/microbench/ab_gemm.sh lib1 lib2 ... ("default" = the built library)
for lib in default "$@"; do
  if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
  echo "== $lib"; python tools/bench_gemm.py 2>&1 | grep "M=" | sed "s/int8 path.*| FP6/FP6/"
done
