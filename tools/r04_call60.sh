#!/bin/bash
# round 4, GPU call 60: ablation builds of the duo kernel (128 x 224): 16 no stores, 8 no DMA after the first, 24 neither, 3 no MFMAs, 4 no quantiser, 7 neither
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c60; mkdir -p $O
for lib in default duoabl16 duoabl8 duoabl24 duoabl3 duoabl4 duoabl7; do
  if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
  echo "== $lib"
  timeout 100 python tools/time_kron.py 128 224 8192 packed f16 2>&1 | grep -v amdgpu.ids
done > $O/time_duo_abl.txt 2>&1
cat $O/time_duo_abl.txt
