#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c8; mkdir -p $O; unset FQHIP_LIB
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
