#!/usr/bin/env python3
"""bench.py — Melem/s of the fused transform + INT4-quant hot path on MI355X.

Default workload (BASELINE.json configs[1], "C2"): Llama-3-8B single linear input, d = 4096 (64 x 64 Kronecker
factors), bs x seq = 8 x 2048 = 16384 tokens PER GPU (weak scaling: tokens shard, no data-path collective;
the two 64x64 factor matrices are broadcast once from rank 0 over RCCL).  One step = one launch of
fq_kron_quant_f16 (packed INT4 + fp16 scales out) over one 128 MiB activation buffer already resident in
HBM; buffers rotate over > 256 MiB so the Infinity Cache cannot serve the stream.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C2S|C1|C3|C4|C4H|C5] [--dtype f16|bf16]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

`--gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset) LAUNCHES the N ranks itself (`launch_ranks`: one process per
GPU under torch.distributed.run on 127.0.0.1 with a free port, RCCL) and fails loudly when the box has fewer than N GPUs; under
an external launcher `--gpus` must equal WORLD_SIZE. Either way stdout is exactly one JSON line with `n_gpus == N`.

--config (the other BASELINE.json configs; one "element" = one input activation scalar of one (token, linear) unit):
  C1  the FlatQuantizedLinear contract of BASELINE configs[0] (flat_linear.py:75-80: per-token W4A4 FAKE-quant of the
      transformed activation, fp16 / bf16 out, 16 KB per token) at C2's size: 64x64 factors, 8 x 2048 tokens per GPU, one
      launch of fq_kron_quant_{f16,bf16}(FQ_OUT_FAKEQUANT | FQ_ROUND_Y_F16) per step. --dtype bf16: the dtype the reference's
      eval pipeline feeds this path on Llama-3 / Qwen / DeepSeek (model_utils.py:20); also valid for C2 (packed out).
  C2S the default workload in its STRONG-scaling form: the 8 x 2048 tokens BASELINE quotes the metric on are split over the
      ranks (2048 tokens per launch at N = 8). At N = 1 identical to C2.
  C3  Llama-3-8B decoder layer, the activation path of its 7 linears: RMSNorm + 64x64 transform with 3 clip sets (q/k/v),
      o_proj head transform (head_dim 128 x 32 heads), RMSNorm + 64x64 with 2 clip sets (up/gate), online Hadamard
      28 x 512 + Quantizer on the down_proj input. 4 launches per step, 8 x 2048 tokens per GPU (weak scaling).
  C4  Llama-2-70B shapes (d = 8192 = 64x128, ffn 28672 = 128x224, 64 heads), all 80 layers per step (320 launches,
      replayed from ONE captured HIP graph), 8 x 2048 tokens in total, rows sharded over the GPUs (strong scaling).
  C4H C4 with the down_proj input of the reference's deploy model under options.trans == "had" (OnlineTrans(had) + Quantizer,
      deploy/transformers/modeling_llama.py:248-253 — BASELINE config 3's "online Hadamard on down_proj" at Llama-2-70B shapes):
      Hadamard 28 x 1024 + Quantizer in one launch instead of the learned 128x224 pair of options.trans == "matmul" (C4).
  C5  DeepSeek-V3 MoE: w1_trans 64x112 over 16384 tokens of d = 7168, then the routed experts' hidden rows
      [8 x 16384, 2048] in 256 groups (Zipf routing) through the grouped 32x64 launch with per-expert clip pairs;
      experts (groups) and tokens sharded over the GPUs (strong scaling).

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (HBM, algorithmic bytes / HIP-
event time per launch of the dominant kernel) and `cpu_baseline` (torch restatement of the reference's CPU fake-quant
path, timed on the host cores, bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

# multi-process GPU work on this platform needs dmabuf IPC (the host driver supports nothing else: without it RCCL fails with
# hipIpcGetMemHandle: invalid argument); the driver's environment exports it already — kept here for a bare shell
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, M, N = 4096, 64, 64
BSZ, SEQ = 8, 2048
ROWS = BSZ * SEQ
BYTES_PER_TOKEN = 2 * D + D // 2 + 2          # fp16 in + packed INT4 out + fp16 scale = 10242
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: 8 TB/s spec
N_BUF = 4                                     # 4 x 128 MiB inputs + 4 x 32 MiB outputs rotate (> 256 MiB L3)
SIG4 = 0.9820137619972229                     # fp32 sigmoid(4.0): the clip factors' initial value


def packed_bytes(d):
    """fp16 in + packed INT4 out + one fp16 scale per token."""
    return 2 * d + d // 2 + 2


def make_inputs(device, seed, rows=ROWS, d=D, n_buf=N_BUF):
    g = torch.Generator(device=device).manual_seed(seed)
    xs = []
    for _ in range(n_buf):
        x = torch.randn(rows, d, generator=g, device=device, dtype=torch.float16)
        x[:, :: 97] *= 20.0                    # LLM-like outlier channels
        xs.append(x)
    return xs


def make_matrix(n, seed, device):
    """random orthogonal . diag(U[0.5, 2]) in fp64 -> fp16."""
    g = torch.Generator().manual_seed(seed)
    q, r = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))
    q = q * torch.sign(torch.diagonal(r))[None, :]
    m = q * (torch.rand(n, generator=g, dtype=torch.float64) * 1.5 + 0.5)[None, :]
    return m.to(torch.float16).contiguous().to(device)


def make_matrices(device):
    return {"left": make_matrix(M, 1, device), "right": make_matrix(N, 2, device)}


def pmc_traffic():
    """(HBM bytes per launch of the dominant kernel, source file) from the COMMITTED rocprofv3 PMC run of this same command
    (tools/prof.sh: separate --pmc passes for FETCH_SIZE and WRITE_SIZE; FETCH_SIZE doubled per the gfx950 correction) — a
    counter pass cannot run inside a timed bench process, so the line names the file the number is read from."""
    for name in ("r06c_kron64_pmc.json", "r06b_kron64_pmc.json", "r06_kron64_pmc.json", "r05_kron64_pmc.json", "r04_kron64_pmc.json", "r03_kron64_pmc.json", "r02_kron64_pmc.json", "r01_kron64_pmc.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                return json.load(fh)["hbm_bytes_per_launch"], "profiles/" + name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """Physical cores of the host (distinct (physical id, core id) pairs of /proc/cpuinfo); logical CPUs when that fails."""
    try:
        seen, phys, core = set(), None, None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_baseline(max_seconds=20.0, m=M, n=N):
    """The reference's CPU fake-quant path (flat_utils.py:6-17 + quant_utils.py:71-119), restated in torch by
    oracle/path_a_torch.py, on a bounded sample: C1-sized batches (2048 tokens) for <= ~20 s. As SURVEY 8d asks: fp16 (the
    dtype BASELINE configs[0] names) AND fp32 (the reference's calibration dtype), each at ONE thread and at ALL PHYSICAL cores
    (torch.set_num_threads), plus a short ladder in between because torch's fp16 CPU GEMM on 64 x 64 factors anti-scales.
    `value` / `cores`: the best fp16 figure; `fp32_value` / `fp32_cores`: the best fp32 one."""
    from oracle import path_a_torch
    rows, d = 2048, m * n
    g = torch.Generator().manual_seed(0)
    x = torch.randn(rows, d, generator=g).to(torch.float16)
    left, right = make_matrix(m, 1, "cpu"), make_matrix(n, 2, "cpu")
    sig = (float(torch.sigmoid(torch.tensor(4.0))),) * 2
    ncpu, nphys = os.cpu_count() or 1, physical_cores()
    ladder = sorted({t for t in (1, 8, 32, nphys) if 1 <= t <= max(nphys, 1)})
    t_start = time.perf_counter()

    def median_rate(xx, ll, rr, budget):
        path_a_torch.kron_fakequant(xx, ll, rr, sig)   # warm-up
        times = []
        t_cfg = time.perf_counter()
        while len(times) < 5 and (not times or time.perf_counter() - t_cfg < budget):
            t0 = time.perf_counter()
            path_a_torch.kron_fakequant(xx, ll, rr, sig)
            times.append(time.perf_counter() - t0)
        times.sort()
        return rows * d / times[len(times) // 2] / 1e6
    results = {"fp16": {}, "fp32": {}}
    operands = {"fp16": (x, left, right), "fp32": (x.float(), left.float(), right.float())}
    # 1 thread and all physical cores first (what SURVEY 8d names), the ladder in between while the budget lasts
    order = [1, nphys] + [t for t in ladder if t not in (1, nphys)]
    per = max_seconds / (2 * max(len(order), 1))
    for threads in order:
        if threads < 1 or (threads not in (1, nphys) and time.perf_counter() - t_start > max_seconds * 0.8):
            continue
        torch.set_num_threads(threads)
        for dt in ("fp16", "fp32"):
            if threads not in results[dt]:
                results[dt][threads] = median_rate(*operands[dt], per)
    best16 = max(results["fp16"], key=results["fp16"].get)
    best32 = max(results["fp32"], key=results["fp32"].get)
    return {"value": results["fp16"][best16], "unit": "Melem/s", "cores": best16, "kind": "port", "cpu_model": cpu_model(),
            "logical_cpus": ncpu, "physical_cores": nphys,
            "fp32_value": results["fp32"][best32], "fp32_cores": best32,
            "sample": f"median of <=5 x ({rows} x {d} tokens, {m}x{n} factors) per dtype and thread count, torch "
                      f"{torch.__version__} CPU ({cpu_model()}, {nphys} physical cores / {ncpu} logical CPUs); value = best fp16 "
                      f"figure, fp32_value = best fp32 figure",
            "by_threads": {str(k): round(v, 2) for k, v in sorted(results["fp16"].items())},
            "fp32_by_threads": {str(k): round(v, 2) for k, v in sorted(results["fp32"].items())}}


# ---------------------------------------------------------------------------------------------------- workloads
class Workload:
    """step(i): one pass of the hot path over one batch (launches on torch's current stream).
    kernels: [(name, fn(i), algorithmic bytes per call)] — the individual launches of a step, timed one by one after the
    timed region to name the dominant kernel. elems: input activation scalars per step on THIS rank."""
    scaling = "weak"
    graph = False
    graph_inputs = 1   # graphs captured when `graph` (one per rotating input set)


class C2(Workload):
    name = "C2"
    metric = "Melems/s for fused kron-transform+INT4-quant, Llama-3-8B d=4096, bs×seq=8×2048"
    fakequant = False      # C1: the fake-quant output (FlatQuantizedLinear's contract) instead of the packed one
    strong = False         # C2S: the same 16384 tokens split over the ranks (strong scaling)

    def __init__(self, device, rank, world, sharding, bcast, dtype="f16"):
        from flatquant_amd import ops
        from flatquant_amd._lib import (FQ_NO_CLAMP0, FQ_OUT_FAKEQUANT, FQ_OUT_PACKED, FQ_ROUND_Y_F16, FQ_WS_PREPARED, check, lib)
        td = torch.bfloat16 if dtype == "bf16" else torch.float16
        mats = make_matrices(device) if rank == 0 else {
            "left": torch.empty(M, M, dtype=torch.float16, device=device),
            "right": torch.empty(N, N, dtype=torch.float16, device=device)}
        mats = bcast(mats)                                     # the only collective on the path (set-up time)
        left, right = mats["left"].to(td).contiguous(), mats["right"].to(td).contiguous()
        sig = [ops.sigmoid_pair(4.0, 4.0)]
        # weak scaling (C1, C2): every rank its own 8 x 2048 tokens; strong scaling (C2S): the 8 x 2048 tokens BASELINE quotes the
        # metric on, split over the ranks (flatquant_amd.sharding.shard_rows) — at N = 8 a rank's launch is 2048 tokens, 5 us
        if self.strong:
            a, b = sharding.shard_rows(ROWS, world, rank)
            ROWS_ = b - a
        else:
            ROWS_ = ROWS
        self.xs = xs = [x.to(td) for x in make_inputs(device, seed=rank, rows=max(ROWS_, 1))]
        fn = lib.fq_kron_quant_bf16 if dtype == "bf16" else lib.fq_kron_quant_f16
        sp = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        # the matrices are constants of a deployed layer: their fragment image is prepared ONCE (fq_kron_prepare_f16, set-up
        # time, like the broadcast) and every launch passes FQ_WS_PREPARED
        wsb = int(lib.fq_kron_workspace_bytes(M, N))
        ws = torch.empty(wsb, dtype=torch.uint8, device=device)
        check((lib.fq_kron_prepare_bf16 if dtype == "bf16" else lib.fq_kron_prepare_f16)(
            left.data_ptr(), right.data_ptr(), M, N, ws.data_ptr(), wsb, sp))
        smax = (ctypes.c_float * 4)(sig[0][0])
        smin = (ctypes.c_float * 4)(sig[0][1])

        def arr(t):
            a = (ctypes.c_void_p * 4)()
            a[0] = t.data_ptr()
            return a
        lp, rp, wp = ctypes.c_void_p(left.data_ptr()), ctypes.c_void_p(right.data_ptr()), ctypes.c_void_p(ws.data_ptr())
        none4 = (ctypes.c_void_p * 4)()
        # pre-allocated rotating outputs; the timed region calls the C ABI directly (no allocator in the loop)
        if self.fakequant:
            flags = FQ_OUT_FAKEQUANT | FQ_ROUND_Y_F16 | FQ_WS_PREPARED      # path A: Y rounded to the activation dtype, fp32 quantiser
            self.fqs = fqs = [torch.empty(max(ROWS_, 1), D, dtype=td, device=device) for _ in range(N_BUF)]
            calls = [(ctypes.c_void_p(xs[i].data_ptr()), arr(fqs[i])) for i in range(N_BUF)]

            def step(i):
                xp, fa = calls[i % N_BUF]
                check(fn(xp, lp, rp, None, ROWS_, M, N, smax, smin, 1, flags, none4, none4, fa, None, wp, wsb, sp))
            bytes_per_token = 4 * D
            what = f"fake-quant {dtype} out (FlatQuantizedLinear._eval_forward contract, 16 KB per token)"
        else:
            flags = FQ_OUT_PACKED | FQ_NO_CLAMP0 | FQ_WS_PREPARED           # deploy OnlineTrans(matmul) contract
            self.qs = qs = [torch.empty(max(ROWS_, 1), D // 2, dtype=torch.uint8, device=device) for _ in range(N_BUF)]
            self.ss = ss = [torch.empty(max(ROWS_, 1), dtype=td, device=device) for _ in range(N_BUF)]
            calls = [(ctypes.c_void_p(xs[i].data_ptr()), arr(qs[i]), arr(ss[i])) for i in range(N_BUF)]

            def step(i):
                xp, qa, sa = calls[i % N_BUF]
                check(fn(xp, lp, rp, None, ROWS_, M, N, smax, smin, 1, flags, qa, sa, none4, None, wp, wsb, sp))
            bytes_per_token = BYTES_PER_TOKEN
            what = f"packed INT4 + {dtype} scale out"
        self._keep = (left, right, ws, smax, smin, calls, none4)
        self.step = step
        self.elems = ROWS_ * D
        self.kernels = [("fq_kron64_kernel", step, ROWS_ * bytes_per_token)]
        self.config = {"workload": f"{self.name}: Llama-3-8B single linear (q_proj input), d=4096 = 64x64 Kronecker, "
                                   f"8x2048 tokens {'in total, rows sharded' if self.strong else 'per GPU'}, {what}",
                       "rows_per_gpu": ROWS_, "d": D, "factors": [M, N], "parallelism": f"rows /{world}" if self.strong else f"rows x{world}",
                       "activation_dtype": dtype, "fragment_image": "prepared once (FQ_WS_PREPARED)"}
        if self.fakequant:
            self.floor_us = None   # (the streaming probe moves the packed contract's byte mix)

    def floor_us(self, stream):
        """practical HBM floor: the same bytes (8 KB in, 2 KB + 2 B out per token) moved by a no-arithmetic kernel"""
        from flatquant_amd import _probe   # (libfqprobe.so: measurement infrastructure, not the product library)
        if self.xs[0].dtype != torch.float16:
            return None
        for i in range(5):
            _probe.probe_stream_4096(self.xs[i % N_BUF], self.qs[i % N_BUF], self.ss[i % N_BUF])
        torch.cuda.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record(stream)
        for i in range(50):
            _probe.probe_stream_4096(self.xs[i % N_BUF], self.qs[i % N_BUF], self.ss[i % N_BUF])
        f1.record(stream)
        torch.cuda.synchronize()
        return f0.elapsed_time(f1) / 50 * 1e3


class C2S(C2):
    """The headline workload in its STRONG-scaling form (missing #6): BASELINE quotes the metric on bs x seq = 8 x 2048
    tokens; here those 16384 tokens are split over the ranks, so the per-rank launch shrinks with N (2048 tokens = 5 us at N = 8:
    the launch, not the kernel, bounds the curve — the honest counterpart of the weak-scaling default)."""
    name = "C2S"
    strong = True
    scaling = "strong"


class C1(C2):
    name = "C1"
    metric = "Melems/s for fused kron-transform+INT4 FAKE-quant (FlatQuantizedLinear contract), Llama-3-8B d=4096, bs×seq=8×2048"
    fakequant = True


def _hadk(K, device):
    from flatquant_amd.flatquant.hadamard_utils import get_hadK
    h, k = get_hadK(K * 512)
    assert k == K
    return h.to(torch.float16).to(device).contiguous()


class C3(Workload):
    name = "C3"
    graph = True        # (round 4) four launches of 33-130 us per step: replayed from a captured HIP graph like C4 / C5, so that the step
    graph_inputs = 2    # is the kernels and not the ~15 us of host work per library call between them; two input sets alternate
    metric = "Melems/s, Llama-3-8B decoder layer activation path (7 linears, W4A4, online Hadamard on down_proj), 8x2048 tokens"

    def __init__(self, device, rank, world, sharding, bcast):
        from flatquant_amd import ops
        from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED
        hid, ffn, H, hd, rows = 4096, 14336, 32, 128, ROWS
        mats = {"ln_l": make_matrix(64, 1, device), "ln_r": make_matrix(64, 2, device), "ug_l": make_matrix(64, 3, device),
                "ug_r": make_matrix(64, 4, device), "o": make_matrix(H, 5, device), "hadk": _hadk(28, device)}
        mats = bcast(mats)
        nb = 2
        xs = make_inputs(device, rank, rows, hid, nb)
        xa = [x.reshape(rows, hd, H) for x in make_inputs(device, rank + 100, rows, hid, nb)]
        x2 = make_inputs(device, rank + 200, rows, hid, nb)
        xf = make_inputs(device, rank + 300, rows, ffn, nb)
        s3, s2, s1 = [(SIG4, SIG4)] * 3, [(SIG4, SIG4)] * 2, (SIG4, SIG4)
        P = FQ_OUT_PACKED | FQ_NO_CLAMP0
        k_qkv = lambda i: ops.rmsnorm_kron_quant(xs[i % nb], 1e-5, mats["ln_l"], mats["ln_r"], s3, P)
        k_o = lambda i: ops.block_quant(xa[i % nb], mats["o"], [s1], P)
        k_ug = lambda i: ops.rmsnorm_kron_quant(x2[i % nb], 1e-5, mats["ug_l"], mats["ug_r"], s2, P)
        k_dn = lambda i: ops.hadamard_quant(xf[i % nb], 28, mats["hadk"], s1)
        self.kernels = [("rmsnorm+kron64 x3 clips (q/k/v)", k_qkv, rows * (2 * hid + 3 * (hid // 2 + 2))),
                        ("block transform (o_proj)", k_o, rows * packed_bytes(hid)),
                        ("rmsnorm+kron64 x2 clips (up/gate)", k_ug, rows * (2 * hid + 2 * (hid // 2 + 2))),
                        ("hadamard 28x512 + Quantizer (down_proj)", k_dn, rows * packed_bytes(ffn))]

        def step(i):
            k_qkv(i), k_o(i), k_ug(i), k_dn(i)
        self.step = step
        self.elems = rows * (3 * hid + hid + 2 * hid + ffn)
        self.config = {"workload": "C3: Llama-3-8B decoder layer, activation path of the 7 linears (RMSNorm+64x64 x3 clips, "
                                   "o_proj head transform 128x32, RMSNorm+64x64 x2 clips, Hadamard 28x512 + Quantizer), "
                                   "8x2048 tokens per GPU", "rows_per_gpu": rows, "launches_per_step": 4,
                       "launch": "HIP graph replay, two input sets alternating",
                       "parallelism": f"rows x{world}"}


class C4(Workload):
    name = "C4"
    down = "kron"       # down_proj input: the learned 128 x 224 pair (C4) | the online Hadamard 28 x 1024 + Quantizer (C4H, below)
    metric = "Melems/s, Llama-2-70B shapes (d=8192, ffn=28672, 64 heads), activation path of all 80 layers, 8x2048 tokens total"
    scaling = "strong"
    graph = True

    def __init__(self, device, rank, world, sharding, bcast):
        from flatquant_amd import ops
        from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED
        hid, ffn, H, hd, layers = 8192, 28672, 64, 128, 80
        a, b = sharding.shard_rows(ROWS, world, rank)
        rows = b - a
        nm = 4                                                   # matrix sets rotate over the layers (per-layer seeds)
        mats = {}
        for j in range(nm):
            mats.update({f"ln_l{j}": make_matrix(64, 10 * j + 1, device), f"ln_r{j}": make_matrix(128, 10 * j + 2, device),
                         f"ug_l{j}": make_matrix(64, 10 * j + 3, device), f"ug_r{j}": make_matrix(128, 10 * j + 4, device),
                         f"o{j}": make_matrix(H, 10 * j + 5, device), f"dn_l{j}": make_matrix(128, 10 * j + 6, device),
                         f"dn_r{j}": make_matrix(224, 10 * j + 7, device)})
        mats = bcast(mats)
        nb = max(2, min(8, (320 << 20) // max(1, rows * ffn * 2)))   # inputs rotate over > 256 MiB
        xs = make_inputs(device, rank, rows, hid, nb)
        xa = [x.reshape(rows, hd, H) for x in make_inputs(device, rank + 100, rows, hid, nb)]
        xf = make_inputs(device, rank + 300, rows, ffn, nb)
        s3, s2, s1 = [(SIG4, SIG4)] * 3, [(SIG4, SIG4)] * 2, [(SIG4, SIG4)]
        P = FQ_OUT_PACKED | FQ_NO_CLAMP0
        # (round 3: deploy.nn.RMSNorm runs inside the 64 x 128 launches, as in C3 — modeling_llama.py:351-357 applies it in
        #  front of both; round 2 had to leave it out because the fusion existed for 64 x 64 only)
        k_qkv = lambda i: ops.rmsnorm_kron_quant(xs[i % nb], 1e-5, mats[f"ln_l{i % nm}"], mats[f"ln_r{i % nm}"], s3, P)
        k_o = lambda i: ops.block_quant(xa[i % nb], mats[f"o{i % nm}"], s1, P)
        k_ug = lambda i: ops.rmsnorm_kron_quant(xs[(i + 1) % nb], 1e-5, mats[f"ug_l{i % nm}"], mats[f"ug_r{i % nm}"], s2, P)
        k_dn = lambda i: ops.kron_quant(xf[i % nb], mats[f"dn_l{i % nm}"], mats[f"dn_r{i % nm}"], s1, P)
        dn_name = "kron 128x224 (down_proj)"
        if self.down == "hadamard":
            hadk = bcast({"hadk": _hadk(28, device)})["hadk"]
            k_dn = lambda i: ops.hadamard_quant(xf[i % nb], 28, hadk, s1[0])
            dn_name = "hadamard 28x1024 + Quantizer (down_proj)"
        self.kernels = [("rmsnorm+kron 64x128 x3 clips (q/k/v)", k_qkv, rows * (2 * hid + 3 * (hid // 2 + 2))),
                        ("block transform 128x64 (o_proj)", k_o, rows * packed_bytes(hid)),
                        ("rmsnorm+kron 64x128 x2 clips (up/gate)", k_ug, rows * (2 * hid + 2 * (hid // 2 + 2))),
                        (dn_name, k_dn, rows * packed_bytes(ffn))]

        def step(i):
            for layer in range(layers):
                k_qkv(layer), k_o(layer), k_ug(layer), k_dn(layer)
        self.step = step
        self.elems = rows * layers * (3 * hid + hid + 2 * hid + ffn)
        self.config = {"workload": "C4: Llama-2-70B shapes, 80 layers x (RMSNorm+64x128 x3 clips, head transform 128x64, "
                                   "RMSNorm+64x128 x2 clips, " + ("128x224" if self.down == "kron" else "Hadamard 28x1024 + Quantizer") +
                                   "), 8x2048 tokens in total, rows sharded; one step = 320 launches replayed "
                                   "from one captured HIP graph", "rows_per_gpu": rows, "layers": layers,
                       "launches_per_step": 4 * layers, "parallelism": f"rows /{world}"}


class C4H(C4):
    """C4 with the down_proj input as the reference's deploy model builds it under options.trans == "had" — OnlineTrans(had) +
    Quantizer in front of Linear4bit (deploy/transformers/modeling_llama.py:248-253; BASELINE config 3's "online Hadamard on down_proj"
    at Llama-2-70B shapes): the rotation of 28672 = 28 x 1024 fused with the Quantizer (the structured kernel, fq_had_mfma.hip) instead
    of the learned 128 x 224 pair of options.trans == "matmul" (C4)."""
    name = "C4H"
    down = "hadamard"


class C5(Workload):
    name = "C5"
    graph = True        # two short launches per step: replayed from a captured HIP graph (host launch cost out of the step)
    graph_inputs = 2
    metric = "Melems/s, DeepSeek-V3 MoE expert inputs: w1_trans 64x112 (d=7168) + grouped 32x64 (2048) over 256 experts, top-8, 16384 tokens"
    scaling = "strong"

    @staticmethod
    def plan(world, rank, sharding):
        """The partition of the workload (host side, no device): routing counts (identical on every rank), this rank's experts
        [e0, e1) with their group offsets over ITS routed rows, and its row block [t0, t1) of the shared w1_trans stage."""
        T, E, K = ROWS, 256, 8
        # routing: seeded multinomial with Zipf-skewed expert popularity, identical on every rank
        g = torch.Generator().manual_seed(5)
        pop = 1.0 / torch.arange(1, E + 1, dtype=torch.float64) ** 0.8
        indices = torch.multinomial(pop[torch.randperm(E, generator=g)].expand(T, E), K, replacement=False, generator=g)
        counts = torch.bincount(indices.flatten(), minlength=E)
        e0, e1, offs = sharding.shard_experts(counts, world, rank)   # EP-style: this rank owns experts [e0, e1) and their rows
        t0, t1 = sharding.shard_rows(T, world, rank)             # and a row block of the shared w1_trans stage
        return {"T": T, "E": E, "K": K, "counts": counts, "e0": e0, "e1": e1, "offs": offs, "t0": t0, "t1": t1}

    def __init__(self, device, rank, world, sharding, bcast, dtype="f16"):
        from flatquant_amd import ops
        from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED
        td = torch.bfloat16 if dtype == "bf16" else torch.float16   # (DeepSeek-V3 runs under torch.set_default_dtype(bfloat16): main_dpskv3.py:395)
        d1, d2 = 7168, 2048
        pl = self.plan(world, rank, sharding)
        T, E, K, counts, e0, e1, offs, t0, t1 = (pl[k] for k in ("T", "E", "K", "counts", "e0", "e1", "offs", "t0", "t1"))
        rows2 = int(offs[-1])
        mats = bcast({"l1": make_matrix(64, 1, device), "r1": make_matrix(112, 2, device),
                      "l2": make_matrix(32, 3, device), "r2": make_matrix(64, 4, device)})
        nb = 2
        x1 = [x.to(td) for x in make_inputs(device, rank, t1 - t0, d1, nb)]
        x2 = [x.to(td) for x in make_inputs(device, rank + 50, rows2, d2, nb)]
        mats = {k: v.to(td) for k, v in mats.items()}
        offs_d = offs.to(device)
        gsig = torch.Generator().manual_seed(6)
        smax = torch.sigmoid(torch.rand(e1 - e0, generator=gsig) * 4 + 1).float().to(device)   # per-expert clip pairs
        smin = torch.sigmoid(torch.rand(e1 - e0, generator=gsig) * 4 + 1).float().to(device)
        P = FQ_OUT_PACKED | FQ_NO_CLAMP0
        k1 = lambda i: ops.kron_quant(x1[i % nb], mats["l1"], mats["r1"], [(SIG4, SIG4)], P)
        k2 = lambda i: ops.kron_quant_grouped(x2[i % nb], mats["l2"], mats["r2"], offs_d, smax, smin, P)
        self.kernels = [("kron 64x112 (w1_trans, all tokens)", k1, (t1 - t0) * packed_bytes(d1)),
                        ("grouped kron 32x64 (routed_w2_trans, per-expert clips)", k2, rows2 * packed_bytes(d2))]

        def step(i):
            k1(i), k2(i)
        self.step = step
        self.elems = (t1 - t0) * d1 + rows2 * d2
        self.config = {"workload": "C5: DeepSeek-V3 MoE, 16384 tokens d=7168 through w1_trans (64x112) + 8 x 16384 routed "
                                   "hidden rows of 2048 in 256 expert groups (Zipf routing) through the grouped 32x64 launch",
                       "tokens_this_rank": t1 - t0, "grouped_rows_this_rank": rows2, "experts_this_rank": e1 - e0,
                       "largest_group": int(counts.max()), "empty_groups": int((counts == 0).sum()),
                       "launches_per_step": 2, "launch": "HIP graph replay, two input sets alternating", "activation_dtype": dtype,
                       "parallelism": f"experts+tokens /{world}"}


class C2SL(Workload):
    """C2S over the LAYERS of the model in one launch (round 4; the `strong` sub-record of the default line): the 8 x 2048 tokens
    BASELINE quotes the metric on are split over the ranks (strong scaling), and a step is the q_proj-input transform of ALL 32
    decoder layers of Llama-3-8B on the rank's shard — 32 independent jobs (own activations, own factor pair) issued as ONE
    multi-job launch (fq_kron_quant_multi_f16), so that at N = 8 a step is one 65536-token launch, not thirty-two 5 us ones."""
    name = "C2S x 32 layers"
    scaling = "strong"
    layers = 32

    def __init__(self, device, rank, world, sharding, bcast):
        from flatquant_amd import ops
        from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED
        a, b = sharding.shard_rows(ROWS, world, rank)
        rows = b - a
        mats = {}
        for j in range(self.layers):
            mats[f"l{j:02d}"], mats[f"r{j:02d}"] = make_matrix(M, 100 + 2 * j, device), make_matrix(N, 101 + 2 * j, device)
        mats = bcast(mats)                                     # one flat broadcast of all layers' matrices (512 KB)
        g = torch.Generator(device=device).manual_seed(1000 + rank)
        self.xs = [torch.randn(max(rows, 1), D, generator=g, device=device, dtype=torch.float16)[:rows] for _ in range(self.layers)]
        self.plan = ops.KronMultiPlan(self.xs, [mats[f"l{j:02d}"] for j in range(self.layers)],
                                      [mats[f"r{j:02d}"] for j in range(self.layers)], (SIG4, SIG4), FQ_OUT_PACKED | FQ_NO_CLAMP0)
        self.step = lambda i: self.plan.run()
        self.elems = self.layers * rows * D
        self.kernels = [("fq_kron64_kernel (multi-job)", self.step, self.layers * rows * BYTES_PER_TOKEN)]
        self.config = {"workload": f"C2S x {self.layers} layers: the 8x2048 tokens split over the ranks, the 64x64 transform + INT4 "
                                   f"quantisation of all {self.layers} layers' q_proj inputs on the rank's shard as ONE multi-job launch",
                       "rows_per_gpu": rows, "layers": self.layers, "launches_per_step": 1, "parallelism": f"rows /{world}"}


WORKLOADS = {"C1": C1, "C2": C2, "C2S": C2S, "C3": C3, "C4": C4, "C4H": C4H, "C5": C5, "C2SL": C2SL}


class TimedBroadcast:
    """sharding.broadcast_matrices with its wall time kept (the one collective of the path, set-up time): .ms after the call."""

    def __init__(self, sharding, device, force=False):
        self.sharding, self.device, self.ms, self.bytes, self.force = sharding, device, 0.0, 0, force

    def __call__(self, mats):
        cuda = torch.device(self.device).type == "cuda"
        if cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = self.sharding.broadcast_matrices(mats, src=0, force=self.force)
        if cuda:
            torch.cuda.synchronize()
        self.ms += (time.perf_counter() - t0) * 1e3
        self.bytes += sum(t.numel() * t.element_size() for t in mats.values())
        return out


def reduce_over_ranks(dist, device, wall_s, kern_ms, elems, force=False):
    """MAX over ranks of the two clocks, SUM of the units, and every rank's own wall clock (all_gather) — the contract's
    'max over ranks' plus what shows a straggler. Works on CPU tensors with gloo (tests) and on the GPU with RCCL."""
    t = torch.tensor([wall_s, kern_ms], dtype=torch.float64, device=device)
    e = torch.tensor([float(elems)], dtype=torch.float64, device=device)
    per_rank = [wall_s]
    if dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        mine = torch.tensor([wall_s], dtype=torch.float64, device=device)
        parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, mine)
        per_rank = [float(p[0]) for p in parts]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(e, op=dist.ReduceOp.SUM)
    return float(t[0]), float(t[1]), float(e[0]), per_rank


def timed_region(step, steps, warmup, dist, stream):
    """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides -> (wall seconds, average step
    in ms from HIP events on the launch stream — of an identical K-step pass right behind the wall-clocked one)."""
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0     # this rank's K steps, start to drained queue; the MAX over ranks is taken by the caller
    if dist is not None:                # the closing barrier + synchronize of the bracket: behind the clock — an RCCL barrier is an
        dist.barrier()                  # all-reduce of its own (~0.2 ms measured in a one-rank group: a third of 20 steps of 35 us)
    torch.cuda.synchronize()
    # The HIP-event figure of the same K steps, from an IDENTICAL pass right behind the wall-clocked one (round 6, third session): an event record
    # is a packet of its own on this runtime — the two records inside the wall-clocked bracket cost it ~0.5 us per step at K = 20
    # (tools/microbench/bracket_probe.py: 34.6 us with them, 34.1 without, events 33.7, K = 1000 33.7) — so the wall clock no longer carries them.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream)                                       # same stream the kernels are launched on
    for i in range(steps):
        step(i)
    ev1.record(stream)
    torch.cuda.synchronize()
    return wall, ev0.elapsed_time(ev1) / steps


def capture_graphs(wl, device):
    """launch-bound step (hundreds of short launches): capture it once, replay it (HIP graph); graph_inputs > 1: one graph per
    rotating input set, replayed in turn, so that a step never re-reads what the previous one left in the Infinity Cache"""
    graphs = []
    for j in range(wl.graph_inputs):
        wl.step(j)
        torch.cuda.synchronize()
        from flatquant_amd import ops
        ops.images_ready()          # the warm-up's fragment images are complete: the captured launches share them (no prepare kernel in the graph)
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            wl.step(j)
        graphs.append(gph)
    return lambda i: graphs[i % len(graphs)].replay()


def sub_record(cls, steps, warmup, device, rank, world, sharding, dist, force=False):
    """One more workload measured inside the same process with the same protocol, reported as a sub-record of the line (round 4:
    the STRONG-scaling forms next to the weak-scaling default, so that a SCALE file cannot be read as 'N x because every GPU got
    its own copy of the work')."""
    bc = TimedBroadcast(sharding, device, force=force)
    wl = cls(device, rank, world, sharding, bc)
    step = capture_graphs(wl, device) if wl.graph else wl.step
    stream = torch.cuda.current_stream(device)
    wall, kern_ms = timed_region(step, steps, warmup, dist, stream)
    wall, kern_ms, elems, per_rank = reduce_over_ranks(dist, device, wall, kern_ms, wl.elems, force)
    rec = None
    if rank == 0:
        step_bytes = sum(k[2] for k in wl.kernels) * (wl.config.get("layers", 1) if issubclass(cls, C4) else 1)
        rec = {"workload": wl.config["workload"], "scaling": wl.scaling, "value": elems / (wall / steps) / 1e6, "unit": "Melem/s",
               "steps": steps, "warmup": warmup, "ms_per_step": wall * 1e3 / steps,
               "per_rank_ms_per_step": [w * 1e3 / steps for w in per_rank], "event_ms_per_step": kern_ms,
               "rows_per_gpu": wl.config.get("rows_per_gpu"), "launches_per_step": wl.config.get("launches_per_step"),
               "broadcast_ms": bc.ms, "broadcast_bytes": bc.bytes,
               "step_hbm_frac_rank0": step_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del wl
    torch.cuda.empty_cache()
    return rec


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, script=None, timeout=None):
    """`bench.py --gpus N` without a launcher around it: start the N ranks ourselves, one process per GPU, the way the reference starts
    its multi-GPU path with an explicit launcher (scripts/deepseek/deepseek-v3/w4a4kv4.sh:7 `torchrun --nproc-per-node ...`,
    main_dpskv3.py:390 reading WORLD_SIZE / RANK / LOCAL_RANK). Re-executes this script under torch.distributed.run on 127.0.0.1 with a
    free port; the children inherit stdout, rank 0 writes the ONE JSON line there. Returns the launcher's exit code (non-zero when any
    rank failed: torch.distributed.run tears the others down)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "8")                    # torch.distributed.run otherwise sets 1 (with a warning): the cpu legs want threads
    return subprocess.run(cmd, env=env, timeout=timeout).returncode


def resolve_world(gpus, environ, device_count):
    """What `--gpus` means given the environment -> ("single" | "ranks" | "launch", world). Raises SystemExit with a clear message when
    the request cannot be honoured: N ranks asked for under a launcher that started a different number, or more GPUs than the box has."""
    ws = environ.get("WORLD_SIZE")
    if ws is not None:                                        # under torch.distributed.run (the driver's N > 1 form) or any launcher
        world = int(ws)
        if gpus is not None and gpus != world:
            raise SystemExit(f"bench.py: --gpus {gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree")
        return ("ranks" if world > 1 else "single"), world
    n = 1 if gpus is None else gpus
    if n < 1:
        raise SystemExit(f"bench.py: --gpus {n}: need at least one GPU")
    if n == 1:
        return "single", 1
    have = device_count()
    if have < n:
        raise SystemExit(f"bench.py: --gpus {n} asked for but this box has {have} GPU(s) visible; refusing to report n_gpus={n} "
                         f"from fewer devices")
    return "launch", n


def dry_run(args, world, rank):
    """--dry-run: the N > 1 plumbing WITHOUT a GPU (tests/test_host_cpu.py drives it with world 2): the launcher, the rank environment,
    the stdout discipline, a gloo group in the place of RCCL, the timed set-up broadcast of the real matrices, the barrier-bracketed
    timed loop (a no-op step), the MAX / SUM reduction and the one JSON line. No kernel runs: `value` is null and the line says so."""
    from flatquant_amd import sharding
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    bc = TimedBroadcast(sharding, "cpu")
    g = torch.Generator().manual_seed(100 + rank)             # only rank 0's values may survive the broadcast
    mats = bc({"left": torch.randn(M, M, generator=g).half(), "right": torch.randn(N, N, generator=g).half()})
    g0 = torch.Generator().manual_seed(100)
    ref = {"left": torch.randn(M, M, generator=g0).half(), "right": torch.randn(N, N, generator=g0).half()}
    same = all(torch.equal(mats[k], ref[k]) for k in ref)
    a, b = sharding.shard_rows(ROWS, world, rank)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    wall = time.perf_counter() - t0 + 1e-9
    if dist is not None:
        dist.barrier()
    wall, _, elems, per_rank = reduce_over_ranks(dist, "cpu", wall, 0.0, (b - a) * D)
    if rank == 0:
        out = {"metric": "dry run: no kernel was launched", "value": None, "unit": "Melem/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "dry_run": True, "broadcast_ok": bool(same), "elems_all_ranks": elems,
               "per_rank_ms_per_step": [w * 1e3 / max(1, args.steps) for w in per_rank],
               "dist": None if dist is None else {"backend": dist.get_backend(), "world_size": dist.get_world_size()}}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one per GPU). N > 1 without WORLD_SIZE in the environment: this process launches the N ranks itself; "
                         "under a launcher it must equal WORLD_SIZE")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: launcher + gloo group + broadcast + reduction + the JSON line (CPU test)")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"], help="activation dtype (C1 / C2 / C5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-records", action="store_true",
                    help="default config only: skip the `strong` (C2S x 32 layers, one multi-job launch) and `c4` sub-records")
    ap.add_argument("--force-dist", action="store_true",
                    help="single process: run inside a ONE-rank RCCL group anyway (init, the set-up broadcast, barriers, the reductions): "
                         "the N > 1 code path on a single-GPU box; the line then carries a `dist` record")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed clock-settle phase before the counted warm-up: launches of the same step for this "
                         "many milliseconds (reported as settle_launches); 0 disables it")
    args = ap.parse_args(argv)
    mode, world = resolve_world(args.gpus, os.environ, (lambda: 1 << 30) if args.dry_run else torch.cuda.device_count)
    if mode == "launch":
        sys.stdout.flush()
        sys.exit(launch_ranks(world, argv))
    if args.steps is None:
        args.steps = {"C1": 500, "C2": 1000, "C2S": 1000, "C3": 100, "C4": 5, "C4H": 5, "C5": 50, "C2SL": 50}[args.config]
    if args.warmup is None:
        args.warmup = {"C1": 100, "C2": 200, "C2S": 200, "C3": 10, "C4": 2, "C4H": 2, "C5": 5, "C2SL": 10}[args.config]
    if args.dtype != "f16" and args.config not in ("C1", "C2", "C5"):
        ap.error("--dtype bf16 goes with --config C1 / C2 / C5 (C3 / C4 are the deploy configs: fp16 contracts of the reference)")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return dry_run(args, world, rank)
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    json_fd = None
    if world > 1 or args.force_dist:
        # ONE JSON line on stdout, and nothing else: RCCL prints a version banner to the C-level stdout at communicator creation (buffered,
        # flushed at exit — i.e. BEHIND the JSON line; seen on the GPU box with RCCL 2.26). Everything written to file descriptor 1 from
        # here on goes to stderr; rank 0 writes its line straight to the saved descriptor at the end.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)   # RCCL
    elif args.force_dist:
        # a ONE-rank RCCL group: init, the set-up broadcast, the barriers and the reductions of the timed region all run on the hardware
        # (no 8-GPU node has been available to any round; this is the part of the N > 1 path a single-GPU box can execute)
        import torch.distributed as dist
        if "MASTER_ADDR" in os.environ and "RANK" in os.environ:
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1, device_id=device)

    from flatquant_amd import sharding
    kw = {"dtype": args.dtype} if args.config in ("C1", "C2", "C2S", "C5") else {}
    bcast = TimedBroadcast(sharding, device, force=args.force_dist)
    wl = WORKLOADS[args.config](device, rank, world, sharding, bcast, **kw)
    stream = torch.cuda.current_stream(device)
    step = wl.step
    if wl.graph:
        step = capture_graphs(wl, device)
        stream = torch.cuda.current_stream(device)

    # Declared, UNTIMED clock-settle phase: the part needs tens of milliseconds of load before its clocks and power
    # state reach steady state (the first ~100 launches of a cold process run 10-20 % slow); a 20-step run otherwise
    # measures the ramp, not the kernel. Not part of --warmup, not part of the timed region; reported below.
    settle_launches = 0
    if args.settle_ms > 0:
        chunk = 64 if args.config in ("C1", "C2", "C2S") else 1
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            for i in range(chunk):
                step(i)
            settle_launches += chunk
            torch.cuda.synchronize()
    wall, kern_ms = timed_region(step, args.steps, args.warmup, dist, stream)

    # Per-step distribution (the reference's own quantiles, deploy/kernels/kron_matmul.py:269-281): a SEPARATE
    # untimed pass of the same K steps with one event pair per step, so that the timed region above carries no
    # event markers between its launches. median / p20 / p80 in microseconds.
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    evs[0].record(stream)
    for i in range(args.steps):
        step(i)
        evs[i + 1].record(stream)
    torch.cuda.synchronize()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(args.steps))
    quant = lambda f: per[min(len(per) - 1, int(f * len(per)))]
    per_launch = {"median_us": quant(0.5), "p20_us": quant(0.2), "p80_us": quant(0.8), "min_us": per[0], "max_us": per[-1]}

    # the individual launches of a step, each timed back to back on its own (HIP events): names the dominant kernel
    kern_us = []
    for name, fn, nbytes in wl.kernels:
        reps = max(10, min(200, args.steps))
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        # (round 6) three windows of `reps` launches, the MEDIAN window: one stall (an allocator growth, a clock step) inside a 10-launch window
        # once read 3479 us for a 123 us kernel (profiles/r06_bench_C4H.json)
        wins = []
        for _ in range(3):
            k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            k0.record(torch.cuda.current_stream(device))
            for i in range(reps):
                fn(i)
            k1.record(torch.cuda.current_stream(device))
            torch.cuda.synchronize()
            wins.append(k0.elapsed_time(k1) / reps * 1e3)
        kern_us.append((name, sorted(wins)[1], nbytes))
    floor_us = wl.floor_us(stream) if (rank == 0 and getattr(wl, "floor_us", None) is not None) else None

    wall, kern_ms, elems_total, per_rank_wall = reduce_over_ranks(dist, device, wall, kern_ms, wl.elems, args.force_dist)

    # (round 4) the STRONG-scaling sub-records of the default line: the driver only ever runs `bench.py --gpus N`, whose headline
    # value is weak scaling (every GPU its own 8 x 2048 tokens)
    subs = {}
    if args.config == "C2" and args.dtype == "f16" and not args.no_sub_records:
        del wl.xs
        torch.cuda.empty_cache()
        if world > 1 or args.force_dist:
            # (round 6) BASELINE quotes the metric on 8 x 2048 tokens IN TOTAL: at N > 1 that exact workload (the tokens split over the ranks,
            # one launch per rank and step) is a first-class record with its own value and per-rank times next to the weak headline —
            # the number that answers ">= 6x at 8 GPUs" (at N = 8 a rank's launch is 2048 tokens = ~5 us: launch-bound, and said so)
            subs["baseline_workload_strong"] = sub_record(C2S, 200, 50, device, rank, world, sharding, dist, args.force_dist)
        subs["strong"] = sub_record(C2SL, 20, 5, device, rank, world, sharding, dist, args.force_dist)
        subs["c4"] = sub_record(C4, 3, 1, device, rank, world, sharding, dist, args.force_dist)

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        value = elems_total / (wall / args.steps) / 1e6
        ev_us = None
        if args.config in ("C1", "C2", "C2S"):               # one launch per step: the timed region IS the kernel
            # (round 6) ONE clock: the roofline figure of the line follows from the same wall clock as `value` and
            # `ms_per_step` (launch gaps included: the kinder event clock of the same region stays beside it as *_events)
            dom_name, dom_us, dom_bytes = wl.kernels[0][0], ms_per_step * 1e3, wl.kernels[0][2]
            ev_us = kern_ms * 1e3
        else:
            dom_name, dom_us, dom_bytes = max(kern_us, key=lambda k: k[1])
        achieved = dom_bytes / (dom_us * 1e-6) / 1e9
        out = {
            "metric": wl.metric, "value": value, "unit": "Melem/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": wl.scaling,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", "settle_launches": settle_launches,
            "settle_ms": args.settle_ms, "config": wl.config,
            "per_rank_ms_per_step": [w * 1e3 / args.steps for w in per_rank_wall],
            "broadcast_ms": bcast.ms, "broadcast_bytes": bcast.bytes,
            "dist": None if dist is None else {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "forced_single_rank": bool(args.force_dist and world == 1)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic()[0] if (args.config == "C2" and args.dtype == "f16") else None,
                         "traffic_source": pmc_traffic()[1] if (args.config == "C2" and args.dtype == "f16") else None,
                         "kernel": dom_name, "algorithmic_bytes_per_launch": dom_bytes,
                         "launch_us": dom_us, "clock": "wall (ms_per_step)" if ev_us is not None else "hip events, the kernel alone back to back",
                         "launch_us_events": ev_us, "achieved_events": (dom_bytes / (ev_us * 1e-6) / 1e9) if ev_us else None,
                         "frac_events": (dom_bytes / (ev_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if ev_us else None,
                         "events_pass": "HIP events around an identical K-step pass right behind the wall-clocked one (an event record is a packet of its own: inside the wall-clocked bracket the two records cost ~0.5 us per step at K = 20)",
                         "per_launch": per_launch, "hbm_stream_floor_us": floor_us,
                         "frac_of_stream_floor": (floor_us / dom_us) if floor_us else None},
        }
        if args.config not in ("C1", "C2"):
            step_bytes = sum(k[2] for k in wl.kernels) * (wl.config.get("layers", 1))
            out["roofline"]["step"] = {"algorithmic_bytes": step_bytes, "step_us": kern_ms * 1e3,
                                       "frac": step_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            out["roofline"]["kernels"] = [{"kernel": n, "launch_us": u, "algorithmic_bytes": b,
                                           "frac": b / (u * 1e-6) / 1e9 / HBM_PEAK_GBS} for n, u, b in kern_us]
        for k, v in subs.items():
            out[k] = v
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(*{"C1": (20.0, 64, 64), "C2": (20.0, 64, 64), "C2S": (20.0, 64, 64), "C3": (20.0, 64, 64),
                                                 "C4": (20.0, 64, 128), "C4H": (20.0, 64, 128), "C5": (20.0, 32, 64),
                                                 "C2SL": (20.0, 64, 64)}[args.config])
        if json_fd is None:
            print(json.dumps(out))
        else:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
