#!/bin/bash
# round 4, GPU call 15: the tiles kernel (80x112, 128x144, 144x192): parity + timing; C3 replayed from a graph; the rmsnorm 3-clip row re-checked
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c15; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kron_tiles.py -x -q > $O/pytest_tiles.txt 2>&1; tail -15 $O/pytest_tiles.txt
timeout 300 python tools/time_kron.py 80 112 16384 packed f16 80 112 16384 packedr f16 128 144 8192 packed f16 128 144 8192 packedr f16 144 192 8192 packed f16 144 192 8192 packedr f16 128 144 16384 packed f16 > $O/time_tiles.txt 2>&1; grep -v amdgpu.ids $O/time_tiles.txt
timeout 300 python bench.py --config C3 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_C3_graph.json 2> $O/bench_C3_graph.err; tail -c 1500 $O/bench_C3_graph.json
timeout 200 python - > $O/rms3.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from flatquant_amd import ops
from flatquant_amd._lib import FQ_OUT_PACKED, FQ_NO_CLAMP0
g = torch.Generator(device="cuda").manual_seed(0)
xs = [torch.randn(16384, 4096, generator=g, device="cuda", dtype=torch.float16) for _ in range(4)]
L = (torch.randn(64, 64, generator=g, device="cuda") / 8).half(); R = (torch.randn(64, 64, generator=g, device="cuda") / 8).half()
def timeit(fn, steps=50, warm=5):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
for name, sig in (("3 equal", [(0.982, 0.982)] * 3), ("3 distinct", [(0.982, 0.982), (0.9, 0.8), (0.7, 0.95)]), ("1", [(0.982, 0.982)]), ("2 equal", [(0.982, 0.982)] * 2)):
    for rep in range(3):
        us = timeit(lambda i: ops.rmsnorm_kron_quant(xs[i % 4], 1e-5, L, R, sig, FQ_OUT_PACKED | FQ_NO_CLAMP0))
        print(f"rmsnorm+kron64 {name:12s} rep {rep}: {us:8.1f} us")
PY
grep -v amdgpu.ids $O/rms3.txt
