for dbg in 0 2048 30720; do echo "DBG=$dbg"; FQ_KRON_DBG=$dbg KRON_ONLY=1 timeout 100 python tools/bench_shapes.py 2>&1 | grep -E "d=14336|d=8192 |d=28672"; done
FQ_KRON_DBG=2048 timeout 100 python -m pytest tests/test_gpu_kron_generic.py -m gpu -x -q 2>&1 | tail -2
