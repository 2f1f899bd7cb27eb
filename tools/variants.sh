#!/bin/bash
# Build measurement variants of libfqhip.so (only ONE source differs: SRC=fq_kron64.hip by default) into variants/ — see tools/time_variants.py.
#   tools/variants.sh name1:"-DFLAG=1 -DOTHER=2" name2:"..."
set -e
cd "$(dirname "$0")/.."
SRC=${SRC:-fq_kron64.hip}
make -C flatquant_amd/csrc -j8 >/dev/null
mkdir -p variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 $flags -c flatquant_amd/csrc/$SRC -o variants/$name.o
  objs=$(ls flatquant_amd/csrc/build/*.o | grep -v "${SRC%.hip}.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libfqhip_$name.so variants/$name.o $objs
  rm variants/$name.o
  echo "built variants/libfqhip_$name.so ($flags)"
done
