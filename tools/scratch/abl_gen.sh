for dbg in 0 2048 0 2048; do echo "DBG=$dbg"; FQ_KRON_DBG=$dbg KRON_ONLY=1 timeout 100 python tools/bench_shapes.py 2>&1 | grep -E "d=14336|d=28672|d=11008"; done
