#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for a in ${ABLS:-0 1 2 4 6 8 16}; do echo "ABLATE=$a"; FQ_GEMM_ABLATE=$a timeout 100 python tools/bench_gemm.py 2>&1 | grep "N=4096 K=4096\|N=4096 K=14336" | cut -c1-150; done
