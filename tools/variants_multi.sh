#!/bin/bash
# Like tools/variants.sh, but SEVERAL sources differ (a macro that lives in a shared header): every file of SRCS is rebuilt with the
# variant's flags (in parallel), the other objects come from flatquant_amd/csrc/build/.
#   SRCS="fq_kron64.hip fq_kron_duo.hip" tools/variants_multi.sh name1:"-DFQ_QUANT_LO=0" name2:"-DFQ_QUANT_LO=2"
set -e
cd "$(dirname "$0")/.."
SRCS=${SRCS:-"fq_block.hip fq_kron64.hip fq_kron_duo.hip fq_kron_tall.hip fq_kron_trio.hip fq_kron_wave.hip fq_kron_generic.hip fq_kron_generic2.hip"}
make -C flatquant_amd/csrc -j8 >/dev/null
mkdir -p variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  mkdir -p variants/obj_$name
  pids=""
  for src in $SRCS; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 $flags \
      -c flatquant_amd/csrc/$src -o variants/obj_$name/${src%.hip}.o &
    pids="$pids $!"
  done
  for p in $pids; do wait $p; done
  objs=""
  for o in flatquant_amd/csrc/build/*.o; do
    b=$(basename $o)
    if [ -f variants/obj_$name/$b ]; then objs="$objs variants/obj_$name/$b"; else objs="$objs $o"; fi
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libfqhip_$name.so $objs
  rm -rf variants/obj_$name
  echo "built variants/libfqhip_$name.so ($flags)"
done
