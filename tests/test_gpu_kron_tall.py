"""GPU parity for fq_kron_tall_kernel (csrc/fq_kron_tall.hip): packed launches with 64 < M <= 192, N = 64 — a wave per ROW
tile, the token never staged in LDS, U exchanged through LDS (172 x 64 = the Hadamard rotation of 11008 as one Kronecker
launch; flatquant_amd/ops.py::_hadamard_as_kron).

The kernel also returns its transform on request, so the quantise + pack stage is checked BIT FOR BIT by applying the
oracle's quantiser to the kernel's own transform (every quantiser route, multi-clip, fp16 Quantizer arithmetic with a
post-scale), the transform itself against the oracle's restatement within the north-star tolerance, and the packed-only
instantiation against the one that also returns the transform; ragged row counts, every row-tile count (MT = 3..6), M % 32
of every kind, grouped launches, repeated full-size launches.
"""
import numpy as np
import pytest
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu
P, F, T, R16, NC0, Q16 = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20
SIG = (0.9820137619972229, 0.9820137619972229)


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def make(M, rows, seed, spike=True):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, M * 64, generator=gen).half()
    if spike and rows:
        x[:, ::97] *= 20
    L = (torch.randn(M, M, generator=gen) / M ** 0.5).half()
    R = (torch.randn(64, 64, generator=gen) / 8.0).half()
    return x.cuda(), L.cuda(), R.cuda()


@pytest.mark.parametrize("M", [65, 80, 96, 97, 112, 128, 129, 140, 160, 161, 172, 192])
@pytest.mark.parametrize("rows", [1, 2, 7, 100, 1500])
def test_quantiser_bit_exact_on_own_transform_and_packed_only_equal(ops, M, rows):
    x, L, R = make(M, rows, M * 7 + rows)
    sigs = [SIG, (0.9, 0.33), (1e-7, 1e-7)]     # magic-number route, clamp route, true-division route
    both = ops.kron_quant(x, L, R, sigs, T | P | R16)           # this kernel, returning its transform too
    y16 = both.y.cpu().numpy().astype(np.float32)
    multi = ops.kron_quant(x, L, R, sigs, P | R16)              # packed only, three clip sets in one launch
    for ci, sig in enumerate(sigs):
        one = ops.kron_quant(x, L, R, [sig], P | R16)
        ref = O.quant_outputs(y16, sig[0], sig[1])
        for o, k in ((one, 0), (multi, ci), (both, ci)):
            assert np.array_equal(o.q[k].cpu().numpy(), ref["packed"]), (M, rows, sig)
            assert np.array_equal(o.scale[k].cpu().numpy(), ref["scale16"]), (M, rows, sig)


@pytest.mark.parametrize("M", [80, 128, 172])
def test_transform_within_tolerance_of_the_oracle(ops, M):
    x, L, R = make(M, 64, M + 3)
    o = ops.kron_quant(x, L, R, [SIG], T | P | R16)
    ref = O.kron_transform(x.cpu().numpy(), L.cpu().numpy(), R.cpu().numpy()).astype(np.float32).reshape(64, -1)
    y = o.y.cpu().numpy().astype(np.float32)
    assert np.all(np.max(np.abs(y - ref), axis=1) <= 1e-3 * np.max(np.abs(ref), axis=1))


@pytest.mark.parametrize("flags", [P | NC0, P, P | R16 | NC0])
def test_flag_routes_packed_only_equals_the_transform_returning_launch(ops, flags):
    x, L, R = make(172, 333, 5)
    x[3] = x[3].abs()                                            # an all-positive token (NO_CLAMP0 matters)
    a = ops.kron_quant(x, L, R, [SIG], flags)
    b = ops.kron_quant(x, L, R, [SIG], flags | T)
    assert torch.equal(a.q[0], b.q[0]) and torch.equal(a.scale[0], b.scale[0])
    if not (flags & R16):       # path B: the fp32 accumulator is quantised; the returned transform is its fp16 rounding
        return
    ref = O.quant_outputs(b.y.cpu().numpy().astype(np.float32), SIG[0], SIG[1], clamp0=not (flags & NC0))
    assert np.array_equal(a.q[0].cpu().numpy(), ref["packed"])


def test_full_size_repeatable(ops):
    """16384 tokens of 172 x 64 (Llama-2-7B ffn width): twenty launches in a row give the same bytes as the launch that
    also returns the transform."""
    x, L, R = make(172, 16384, 11)
    ref = ops.kron_quant(x, L, R, [SIG], P | T | NC0)
    q0, s0 = ref.q[0].clone(), ref.scale[0].clone()
    del ref
    for _ in range(20):
        o = ops.kron_quant(x, L, R, [SIG], P | NC0)
        assert torch.equal(o.q[0], q0) and torch.equal(o.scale[0], s0)


def test_grouped_launch(ops):
    """Per-group clip pairs (fq_kron_quant_grouped_f16) through this kernel: equal to one launch per group."""
    x, L, R = make(140, 700, 17)
    offs = torch.tensor([0, 0, 5, 5, 260, 699, 700], dtype=torch.int64, device="cuda")   # empty groups, a 1-token group
    G = offs.numel() - 1
    gen = torch.Generator().manual_seed(1)
    smax = (0.5 + 0.5 * torch.rand(G, generator=gen)).cuda()
    smin = (0.3 + 0.7 * torch.rand(G, generator=gen)).cuda()
    o = ops.kron_quant_grouped(x, L, R, offs, smax, smin, P | NC0)
    for g in range(G):
        a, b = int(offs[g]), int(offs[g + 1])
        if a == b:
            continue
        one = ops.kron_quant(x[a:b].contiguous(), L, R, [(float(smax[g]), float(smin[g]))], P | NC0)
        assert torch.equal(o.q[0][a:b], one.q[0]) and torch.equal(o.scale[0][a:b], one.scale[0]), g


@pytest.mark.parametrize("M", [96, 172])
def test_fp16_quantiser_with_post_scale(ops, M):
    """fq_kron_quant_ex_f16 with the deploy Quantizer's fp16 arithmetic and a post-scale (a Hadamard rotation as a Kronecker
    pair): the quantiser bit-exact on the transform the same launch returns, packed-only launch identical."""
    from flatquant_amd._lib import FQ_SIG_F16
    x, L, R = make(M, 257, 23, spike=False)
    ps = 1.0 / 3.0
    for sig, extra in (((1.0, 1.0), 0), (SIG, FQ_SIG_F16)):
        both = ops.kron_quant_ex(x, L, R, ps, [sig], T | P | R16 | Q16 | extra)
        y = both.y.cpu().numpy().astype(np.float32)
        ref = O.quant_outputs(y, sig[0], sig[1], quant_f16=True, sig_f16=bool(extra))
        o = ops.kron_quant_ex(x, L, R, ps, [sig], P | R16 | Q16 | extra)
        for r in (both, o):
            assert np.array_equal(r.scale[0].cpu().numpy(), ref["scale16"])
            assert np.array_equal(r.q[0].cpu().numpy(), ref["packed"])
