#!/bin/bash
# round 4, GPU call 4: the two-operation fp16 quotient (exhaustive), the whole parity suite on the new build, timings of the fp16-epilogue launches
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c4; mkdir -p $O
timeout 300 tools/scratch/h16div2 > $O/h16div2.txt 2>&1; cat $O/h16div2.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
cat > /tmp/time_had.py <<'PY'
import torch, statistics, sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from flatquant_amd import ops
from flatquant_amd._lib import FQ_OUT_PACKED, FQ_QUANT_F16, FQ_SIG_F16
from tests.conftest import hadk_matrix
rows = 16384
g = torch.Generator(device="cuda").manual_seed(0)
sig = (0.9820137619972229, 0.9820137619972229)
def timeit(f, steps=50, rounds=5):
    for i in range(10): f(i)
    torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps): f(i)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / steps * 1e3)
    return statistics.median(ts), min(ts)
for n, K in ((14336, 28), (11008, 172), (28672, 28)):
    r = rows if n < 20000 else rows // 2
    xs = [torch.randn(r, n, generator=g, device="cuda", dtype=torch.float32).half() for _ in range(2)]
    hk = torch.from_numpy(hadk_matrix(K)).cuda()
    pb = r * (2.5 * n + 2)
    cases = [("hadamard_quant default", lambda i: ops.hadamard_quant(xs[i % 2], K, hk, sig), pb),
             ("hadamard_quant kron (dense)", lambda i: ops.hadamard_quant(xs[i % 2], K, hk, sig, route="kron"), pb),
             ("hadamard_quant fwht", lambda i: ops.hadamard_quant(xs[i % 2], K, hk, sig, route="fwht"), pb),
             ("hadamard default (fp16 out)", lambda i: ops.hadamard(xs[i % 2], K, hk), r * 4.0 * n),
             ("hadamard fwht (fp16 out)", lambda i: ops.hadamard(xs[i % 2], K, hk, fwht_route=True), r * 4.0 * n),
             ("deploy Quantizer (rowquant fp16)", lambda i: ops.rowquant(xs[i % 2], [sig], FQ_OUT_PACKED | FQ_QUANT_F16 | FQ_SIG_F16), pb)]
    for name, f, b in cases:
        us, mn = timeit(f)
        print(f"n={n:5d} rows={r:5d} {name:34s} {us:8.1f} us (min {mn:.1f})  {b / us / 1e3:7.0f} GB/s  {b / us / 8e6:5.3f} of 8 TB/s", flush=True)
    del xs
PY
for rep in 1 2; do
  for lib in default g4; do
    if [ "$lib" = default ]; then unset FQHIP_LIB; else export FQHIP_LIB=variants/libfqhip_$lib.so; fi
    echo "== $lib"

    timeout 300 python /tmp/time_had.py
  done
done > $O/time_had.txt 2>&1
cat $O/time_had.txt | grep -v amdgpu.ids
