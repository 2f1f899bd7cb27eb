"""The transform as the GEMM's prologue, decode regime (fq_kron64_linear_multi_f16, round 6): ONE launch for ln_trans -> quantizer -> the
q / k / v (or up / gate) projections of 1..16 tokens (deploy/transformers/modeling_llama.py:66-78,268-280; deploy/nn/linear.py:40-54).
The bar is bit-exactness against the two launches it replaces — which tests/test_gpu_kron64.py, test_gpu_rmsnorm.py and test_gpu_gemm.py
hold to the oracle and the reference's goldens — and, directly, against the oracle's own chain."""
import numpy as np
import pytest
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def _mats(gen, n):
    a = torch.randn(n, n, generator=gen) / n ** 0.5 + torch.eye(n) * 0.5
    return a.half().cuda()


def _problem(ops, gen, N, bias):
    wq = torch.randint(-8, 8, (N, 4096), generator=gen, dtype=torch.int32).numpy()
    wp = torch.from_numpy(O.pack_i4(wq)).cuda()
    ws = (torch.rand(N, generator=gen) * 0.01 + 0.001).half().cuda()
    b = (torch.randn(N, generator=gen) * 0.1).half().cuda() if bias else None
    return wp, ops.int4_to_frag(wp), ws, b


@pytest.mark.parametrize("M", [1, 2, 7, 8, 9, 16])
@pytest.mark.parametrize("eps", [None, 1e-5])
def test_fused_launch_equals_the_two_launches(ops, M, eps):
    from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED
    gen = torch.Generator().manual_seed(100 * M + (1 if eps else 0))
    x = (torch.randn(M, 4096, generator=gen) * (1 + 3 * torch.rand(M, 1, generator=gen))).half().cuda()
    if M > 2:
        x[1] = 0                # an all-zero token
        x[2, 99] = 1500.0       # an outlier: the other digits are 0 / +-1
    L, R = _mats(gen, 64), _mats(gen, 64)
    # q / k / v shapes (1 tile per workgroup) and an up / gate pair wide enough that a workgroup walks several tiles with the ring
    for Ns, biases in (((4096, 1024, 1024), (False, True, False)), ((14336, 14336), (False, False)), ((32,), (True,)), ((9632, 64, 4096, 320), (True, False, False, True))):
        sigs = [(1.0, 1.0), (0.71, 0.83), (0.5, 0.45), (0.93, 0.6)][:len(Ns)]
        probs = [_problem(ops, gen, N, b) for N, b in zip(Ns, biases)]
        for fl in (0, FQ_NO_CLAMP0):
            ref_q = (ops.rmsnorm_kron_quant(x, eps, L, R, sigs, FQ_OUT_PACKED | fl) if eps is not None
                     else ops.kron_quant(x, L, R, sigs, FQ_OUT_PACKED | fl))
            ref = ops.int4_skinny_linear_multi([(ref_q.q[p], ref_q.scale[p], probs[p][1], probs[p][2], probs[p][3]) for p in range(len(Ns))])
            got = ops.kron64_linear_multi(x, L, R, sigs, [(pr[1], pr[2], pr[3]) for pr in probs], eps=eps, flags=fl)
            assert got is not None
            for p in range(len(Ns)):
                assert torch.equal(ref[p], got[p]), (Ns, p, fl, int((ref[p] != got[p]).sum()))


def test_fused_launch_against_the_oracle_chain(ops):
    """x -> kron_quant (oracle) -> linear4bit (oracle), 5 tokens, two problems"""
    gen = torch.Generator().manual_seed(3)
    M = 5
    x = torch.randn(M, 4096, generator=gen).half()
    L, R = _mats(gen, 64), _mats(gen, 64)
    Ns, sigs = (256, 96), [(0.9, 0.8), (1.0, 1.0)]
    probs = [_problem(ops, gen, N, True) for N in Ns]
    got = ops.kron64_linear_multi(x.cuda(), L, R, sigs, [(pr[1], pr[2], pr[3]) for pr in probs])
    for p, N in enumerate(Ns):
        o = O.kron_quant(x.numpy(), L.cpu().numpy(), R.cpu().numpy(), sig_max=sigs[p][0], sig_min=sigs[p][1])
        y = O.linear4bit(o["packed"], o["scale16"], probs[p][0].cpu().numpy(), probs[p][2].cpu().numpy(), probs[p][3].cpu().numpy())
        g = got[p].cpu().numpy()
        # the oracle's transform accumulates in another order than the MFMA: a digit may flip on a rounding tie (the 1e-3 bar of the
        # packed suites); compare the outputs of the tokens whose digits agree with the kernel's own packed launch instead
        from flatquant_amd._lib import FQ_OUT_PACKED
        kq = ops.kron_quant(x.cuda(), L, R, [sigs[p]], FQ_OUT_PACKED)
        same = (O.unpack_i4(kq.q[0].cpu().numpy()) == o["q"]).all(axis=1)
        assert same.sum() >= M - 1
        assert np.array_equal(g[same].view(np.uint16), y[same].view(np.uint16))


def test_uncovered_shapes_return_none_or_say_so(ops):
    gen = torch.Generator().manual_seed(4)
    L, R = _mats(gen, 64), _mats(gen, 64)
    pr = _problem(ops, gen, 64, False)
    x17 = torch.randn(17, 4096, generator=gen).half().cuda()
    assert ops.kron64_linear_multi(x17, L, R, [(1.0, 1.0)], [(pr[1], pr[2], pr[3])]) is None
    from flatquant_amd import _lib
    import ctypes
    y = torch.empty(17, 64, dtype=torch.float16, device="cuda")
    ws = torch.empty(32768, dtype=torch.uint8, device="cuda")
    one = (ctypes.c_float * 4)(1.0, 1.0, 1.0, 1.0)
    VP = ctypes.c_void_p * 1
    rc = _lib.lib.fq_kron64_linear_multi_f16(x17.data_ptr(), 0, ctypes.c_float(0.0), L.data_ptr(), R.data_ptr(), 17, 1, one, one, 0,
                                             VP(pr[1].data_ptr()), VP(pr[2].data_ptr()), VP(None), (ctypes.c_int * 1)(64), VP(y.data_ptr()),
                                             ws.data_ptr(), 32768, None)
    assert rc == _lib.FQ_EUNSUPPORTED


@pytest.mark.parametrize("Ns", [(40960,), (14336, 14336, 14336, 14336), (4096, 32, 22528)])
def test_long_tile_sequences_per_workgroup(ops, Ns):
    """five to seven feature tiles per workgroup: the steady-state iterations of the two-slot weight ring (every wait count of the loop, odd and
    even tile counts, the last tile in either slot), against the two launches"""
    from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED
    gen = torch.Generator().manual_seed(sum(Ns))
    M = 3
    x = torch.randn(M, 4096, generator=gen).half().cuda()
    L, R = _mats(gen, 64), _mats(gen, 64)
    sigs = [(1.0, 1.0), (0.71, 0.83), (0.5, 0.45), (0.93, 0.6)][:len(Ns)]
    probs = [_problem(ops, gen, N, p == 0) for p, N in enumerate(Ns)]
    ref_q = ops.rmsnorm_kron_quant(x, 1e-6, L, R, sigs, FQ_OUT_PACKED | FQ_NO_CLAMP0)
    ref = ops.int4_skinny_linear_multi([(ref_q.q[p], ref_q.scale[p], probs[p][1], probs[p][2], probs[p][3]) for p in range(len(Ns))])
    for _ in range(3):          # (repeatable: no state survives a launch)
        got = ops.kron64_linear_multi(x, L, R, sigs, [(pr[1], pr[2], pr[3]) for pr in probs], eps=1e-6, flags=FQ_NO_CLAMP0)
        for p in range(len(Ns)):
            assert torch.equal(ref[p], got[p]), (Ns, p, int((ref[p] != got[p]).sum()))
