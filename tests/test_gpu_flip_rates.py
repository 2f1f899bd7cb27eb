"""SURVEY 7's bar for routes that share the arithmetic up to rounding noise — INT4 digits within +-1 on <= 1e-3 of the elements — held on
populations of >= 1e6 digits (a dozen small-sample sites assert 2e-3 on 37-row tensors). Every route those sites cover:
the fused kernels against the oracle pipeline, the RMSNorm fusions against the un-fused pair of launches, the matrix-pipe Hadamard routes
against the FWHT route. Rates are recorded (gpurun_out/flip_rates.txt -> profiles/r06_flip_rates.txt)."""
import numpy as np
import pytest
import torch

from conftest import flip_ok, hadk_matrix
from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu
P, T, R16, NC0, QF16, SF16 = 0x01, 0x04, 0x08, 0x10, 0x20, 0x400
SIGS = [(0.982, 0.953), (0.7, 0.9), (1.0, 1.0)]


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def llm_like(rng, rows, d):
    x = (rng.standard_normal((rows, d)) * rng.uniform(0.05, 20, (rows, 1))).astype(np.float16)
    x[:, ::97] *= 8
    return x


@pytest.mark.parametrize("M,N,rows", [(64, 64, 512), (64, 128, 256), (64, 112, 256), (112, 128, 128), (128, 224, 64), (86, 128, 128),
                                      (32, 64, 1024), (128, 148, 64), (172, 64, 128)])
def test_fused_kernels_vs_oracle_pipeline(ops, M, N, rows):
    rng = np.random.default_rng(M * 1000 + N)
    x = llm_like(rng, rows, M * N)
    L = (rng.standard_normal((M, M)) / np.sqrt(M)).astype(np.float16)
    R = (rng.standard_normal((N, N)) / np.sqrt(N)).astype(np.float16)
    o = ops.kron_quant(torch.from_numpy(x).cuda(), torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), SIGS[:2], P | NC0)
    for ci, (a, b) in enumerate(SIGS[:2]):
        ref = O.kron_quant(x, L, R, a, b, clamp0=False)
        q = O.unpack_i4(o.q[ci].cpu().numpy())
        assert q.size >= 1_000_000
        assert flip_ok(q, ref["q"].astype(np.int32), f"[1e6] kron {M}x{N} clip {ci} vs oracle pipeline"), (M, N, ci)


@pytest.mark.parametrize("M,N", [(64, 64), (64, 128), (64, 112), (32, 64)])
def test_rmsnorm_fusions_vs_two_launches_and_oracle(ops, M, N):
    rng = np.random.default_rng(M * 77 + N)
    rows = max(256, (1 << 20) // (M * N))
    x = llm_like(rng, rows, M * N)
    L = (rng.standard_normal((M, M)) / np.sqrt(M)).astype(np.float16)
    R = (rng.standard_normal((N, N)) / np.sqrt(N)).astype(np.float16)
    xd, Ld, Rd = torch.from_numpy(x).cuda(), torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    fused = ops.rmsnorm_kron_quant(xd, 1e-5, Ld, Rd, SIGS, P | NC0)
    two = ops.kron_quant(ops.rmsnorm(xd, 1e-5), Ld, Rd, SIGS, P | NC0)
    xn = O.rmsnorm(x, 1e-5)
    for ci, (a, b) in enumerate(SIGS):
        qf, q2 = O.unpack_i4(fused.q[ci].cpu().numpy()), O.unpack_i4(two.q[ci].cpu().numpy())
        assert qf.size >= 1_000_000
        assert flip_ok(qf, q2, f"[1e6] rmsnorm + {M}x{N} clip {ci} vs two launches"), (M, N, ci)
        if ci == 0:
            ref = O.kron_quant(xn, L, R, a, b, clamp0=False)
            assert flip_ok(qf, ref["q"].astype(np.int32), f"[1e6] rmsnorm + {M}x{N} vs oracle pipeline"), (M, N)


@pytest.mark.parametrize("n,K,rows", [(14336, 28, 128), (28672, 28, 64), (11008, 172, 128), (8960, 140, 128)])
def test_matrix_pipe_hadamard_routes_vs_fwht_route(ops, n, K, rows):
    g = torch.Generator().manual_seed(n + K)
    x = torch.randn(rows, n, generator=g).half()
    x[:, ::53] *= 12
    x = x.cuda()
    hk = torch.from_numpy(hadk_matrix(K)).cuda()
    for sig in SIGS[:2]:
        q, s = ops.hadamard_quant(x, K, hk, sig)
        two = ops.rowquant(ops.hadamard(x, K, hk, fwht_route=True), [sig], P | QF16 | SF16)          # = deploy Quantizer behind the FWHT kernel
        qa, qb = O.unpack_i4(q.cpu().numpy().reshape(rows, -1)), O.unpack_i4(two.q[0].cpu().numpy())
        assert qa.size >= 1_000_000
        assert flip_ok(qa, qb, f"[1e6] hadamard_quant n={n} K={K} sig={sig[0]:.2f} vs FWHT + Quantizer"), (n, K, sig)
