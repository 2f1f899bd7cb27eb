#!/bin/bash
# ablation timings of fq_kron_duo_kernel (variants built with SRC=fq_kron_duo.hip tools/variants.sh name:"-DDUO_ABL=n")
for lib in default "$@"; do
  if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
  echo "== $lib"; python tools/time_kron.py 128 224 8192 packed f16
done
