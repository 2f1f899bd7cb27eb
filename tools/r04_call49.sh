#!/bin/bash
# round 4, GPU call 49: gate / up epilogue with the read-back requested in front of the next tile's LDS-DMA requests: parity, timing, ablations
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c49; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gemm_bf6.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for v in default gu1 gu3; do
  echo "== $v" >> $O/time.txt
  if [ $v = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=$PWD/variants/libfqhip_$v.so; fi
  timeout 300 python tools/time_gate_up.py 2>&1 | grep -v amdgpu.ids >> $O/time.txt
done
cat $O/time.txt
