#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
mkdir -p gpurun_out/r04c12
timeout 300 python tools/host_overhead.py --profile > gpurun_out/r04c12/host.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04c12/host.txt | head -90
