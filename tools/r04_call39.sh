#!/bin/bash
# round 4, GPU call 39: wave kernel with fake-quant / transform outputs: parity, timing against the workgroup-per-token kernel (round-4 table)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c40; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kron_generic.py tests/test_gpu_bf16.py tests/test_gpu_round2.py tests/test_gpu_group128.py -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 300 python tools/time_kron.py 64 128 16384 fq f16 64 128 16384 y f16 64 128 16384 fq bf16 64 112 16384 fq f16 64 112 16384 y f16 32 64 16384 fq f16 32 64 16384 y f16 32 64 16384 fq bf16 64 80 16384 fq f16 56 64 16384 fq f16 2>&1 | grep -v amdgpu.ids > $O/time.txt; cat $O/time.txt
