#!/bin/bash
# Build measurement variants of ONE kernel file (SRC=fq_kron64.hip by default) as small OVERLAY objects variants/ov_<name>.so:
#   tools/variants.sh name1:"-DFLAG=1 -DOTHER=2" name2:"..."
#   FQHIP_OVERLAY=variants/ov_name1.so python tools/time_kron.py ...
# flatquant_amd/_lib.py loads the overlay RTLD_GLOBAL in front of libfqhip.so; the overlay's fq_launch_* functions replace the library's
# (~100 KB - 1 MB per variant; the full-library copies of rounds 2-4 were 13 MB each and dominated the gpurun push).
# fq_gemm_bf6.hip variants need the file's own extra flag: pass "-mllvm -disable-machine-sink" in the spec.
set -e
cd "$(dirname "$0")/.."
SRC=${SRC:-fq_kron64.hip}
mkdir -p variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  [ "$flags" = "$spec" ] && flags=""
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 $flags \
        -shared -o variants/ov_$name.so flatquant_amd/csrc/$SRC
  echo "built variants/ov_$name.so ($SRC $flags)"
done
