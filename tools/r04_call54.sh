#!/bin/bash
# round 4, GPU call 54: ablation builds of the structured Hadamard kernel (HM_ABL: 16 no DMA after the first, 8 no stores, 24 neither, 1 no quantiser,
# 6 no GEMMs, 96 no butterflies) — where the 130 / 145 us go
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c54; mkdir -p $O
export TIME_HAD_FAST=1
for lib in default hmabl16 hmabl8 hmabl24 hmabl1 hmabl6 hmabl96; do
  if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
  echo "== $lib"
  timeout 100 python tools/time_had.py 28672:28 14336:28 2>&1 | grep -v amdgpu.ids
done > $O/time_had_abl.txt 2>&1
cat $O/time_had_abl.txt
