#!/bin/bash
# round 4, GPU call 48: where the gate / up epilogue's time goes: measurement builds without the SiLU arithmetic (gu1), without the read-back (gu2), without both (gu3)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c48; mkdir -p $O
for v in default gu1 gu2 gu3; do
  echo "== $v" >> $O/time.txt
  if [ $v = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=$PWD/variants/libfqhip_$v.so; fi
  timeout 300 python tools/time_gate_up.py 2>&1 | grep -v amdgpu.ids >> $O/time.txt
done
cat $O/time.txt
