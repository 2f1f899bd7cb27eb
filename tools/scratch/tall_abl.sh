#!/bin/bash
# ablation timings of fq_kron_tall_kernel (variants built with SRC=fq_kron_tall.hip tools/variants.sh name:"-DTALL_ABL=n")
for lib in default "$@"; do
  if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
  echo "== $lib"; python tools/time_kron.py 172 64 16384 packed f16 172 64 16384 h16 f16 140 64 16384 packed f16 140 64 16384 h16 f16 96 64 16384 h16 f16
done
