#!/bin/bash
# A/B builds prepared at the end of round 4 (no GPU budget was left to time them). On the container:
#   SRC=fq_had_mfma.hip tools/variants.sh hmnt:"-DHM_COPY_NT=1" hmlate:"-DHM_COPY_LATE=1" hmntlate:"-DHM_COPY_NT=1 -DHM_COPY_LATE=1" hmystage:"-DHM_YSTAGE=1"
# (hmystage: the rotation-only launches with the rotated token staged through the group's own buffer — written but never run on a GPU: run its parity first,
#  FQHIP_LIB=variants/libfqhip_hmystage.so python -m pytest tests/test_gpu_had_mfma.py tests/test_gpu_hadamard.py -q -m gpu)
# then, through gpurun (about 40 s of box time):
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/next_ab; mkdir -p $O
export TIME_HAD_FAST=1
for rep in 1 2; do
for lib in default hmnt hmlate hmntlate hmystage; do
  if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
  [ "$lib" = default ] || [ -f "$FQHIP_LIB" ] || continue
  echo "== $lib"
  timeout 100 python tools/time_had.py 14336:28 28672:28 2>&1 | grep -v amdgpu.ids
done
done > $O/time_had.txt 2>&1
cat $O/time_had.txt
# parity of a variant before it becomes the default: FQHIP_LIB=variants/libfqhip_<name>.so python -m pytest tests/test_gpu_had_mfma.py -q -m gpu
