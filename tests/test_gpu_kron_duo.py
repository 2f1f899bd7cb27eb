"""GPU parity for fq_kron_duo_kernel (csrc/fq_kron_duo.hip): packed-only launches with 96 < M <= 128, N = 224
(28672 = 128 x 224, Llama-2-70B down_proj — BASELINE config 4's dominant launch).

Like the trio kernel it shares only the fragment workspace and the quantiser helpers with the workgroup-per-token kernel:
token staging, the streamed R fragments, its LDS-counter meetings, its token claims and its stores are its own. Every case
is compared BIT FOR BIT with the workgroup-per-token kernel (a launch that also asks for the transform takes that one) and
with the oracle's quantiser on that transform, over ragged row counts (fewer tokens than groups, odd counts, more workgroups
than CUs), every kind of M the kernel admits, grouped launches and repeated launches.
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
    x = torch.randn(rows, M * 224, generator=gen).half()
    if spike and rows:
        x[:, ::97] *= 20
    L = (torch.randn(M, M, generator=gen) / M ** 0.5).half()
    R = (torch.randn(224, 224, generator=gen) / 224 ** 0.5).half()
    return x.cuda(), L.cuda(), R.cuda()


@pytest.mark.parametrize("M", [97, 100, 112, 120, 127, 128])
@pytest.mark.parametrize("rows", [1, 2, 3, 7, 100, 777])
def test_bit_equal_to_workgroup_kernel_and_oracle(ops, M, rows):
    x, L, R = make(M, rows, M * 7 + rows)
    sigs = [SIG, (0.9, 0.33), (1e-7, 1e-7)]     # magic-number route, clamp route, true-division route
    both = ops.kron_quant(x, L, R, sigs, T | P | R16)           # workgroup-per-token kernel (asks for the transform too)
    y16 = both.y.cpu().numpy().astype(np.float32)
    multi = ops.kron_quant(x, L, R, sigs, P | R16)              # this kernel, three clip sets in one launch
    for ci, sig in enumerate(sigs):
        one = ops.kron_quant(x, L, R, [sig], P | R16)           # this kernel, one clip set
        ref = O.quant_outputs(y16, sig[0], sig[1])
        for o, k in ((one, 0), (multi, ci)):
            assert torch.equal(o.q[k], both.q[ci]), (M, rows, sig)
            assert torch.equal(o.scale[k], both.scale[ci]), (M, rows, sig)
            assert np.array_equal(o.q[k].cpu().numpy(), ref["packed"]), (M, rows, sig)
            assert np.array_equal(o.scale[k].cpu().numpy(), ref["scale16"]), (M, rows, sig)


@pytest.mark.parametrize("flags", [P | NC0, P, P | R16 | NC0])
def test_flag_routes_bit_equal_to_workgroup_kernel(ops, flags):
    """Path A / path B rounding and the no-clamp statistics: same bits as the kernel that also returns the transform."""
    x, L, R = make(128, 333, 5)
    x[3] = x[3].abs()                                            # an all-positive token (NO_CLAMP0 matters)
    a = ops.kron_quant(x, L, R, [SIG], flags)
    b = ops.kron_quant(x, L, R, [SIG], flags | T)
    assert torch.equal(a.q[0], b.q[0]) and torch.equal(a.scale[0], b.scale[0])


def test_full_size_bit_equal_and_repeatable(ops):
    """BASELINE config 4 size (8192 tokens of 128 x 224): every token equals the workgroup-per-token kernel's, and twenty
    launches in a row give the same bytes (tokens are claimed dynamically: the schedule differs from launch to launch)."""
    x, L, R = make(128, 8192, 11)
    ref = ops.kron_quant(x, L, R, [SIG], P | T | NC0)
    q0, s0 = ref.q[0].clone(), ref.scale[0].clone()
    del ref
    for _ in range(20):
        o = ops.kron_quant(x, L, R, [SIG], P | NC0)
        assert torch.equal(o.q[0], q0) and torch.equal(o.scale[0], s0)


def test_grouped_launch(ops):
    """Per-expert clip pairs (fq_kron_quant_grouped_f16) through this kernel: equal to one launch per group."""
    x, L, R = make(128, 700, 17)
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


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("M", [128, 120, 100])
@pytest.mark.parametrize("rows", [1, 7, 300])
def test_bf16_packed_only_launches(ops, M, rows):
    """bf16 activations and factors on this kernel (round 4; they ran the workgroup-per-token kernel: 254 us per 8192 tokens): bit-equal,
    clip set by clip set and flag route by flag route, to the launch that also returns the transform, and to the oracle's bf16
    quantiser on that transform."""
    BF = torch.bfloat16
    gen = torch.Generator().manual_seed(M * 31 + rows)
    x = torch.randn(rows, M * 224, generator=gen)
    x[:, ::97] *= 20
    x = x.to(BF).cuda()
    L = (torch.randn(M, M, generator=gen) / M ** 0.5).to(BF).cuda()
    R = (torch.randn(224, 224, generator=gen) / 224 ** 0.5).to(BF).cuda()
    sigs = [SIG, (0.9, 0.33), (1e-7, 1e-7)]
    for flags in (P | NC0, P, P | R16 | NC0):
        both = ops.kron_quant(x, L, R, sigs, flags | T)
        multi = ops.kron_quant(x, L, R, sigs, flags)
        for ci, sig in enumerate(sigs):
            one = ops.kron_quant(x, L, R, [sig], flags)
            for o, k in ((one, 0), (multi, ci)):
                assert torch.equal(o.q[k], both.q[ci]), (M, rows, flags, sig)
                assert np.array_equal(_bits(o.scale[k]), _bits(both.scale[ci])), (M, rows, flags, sig)
        if flags & R16:
            ref = O.quant_outputs(O.bf16_from_bits(_bits(both.y)), *sigs[0], round_y_f16=True, clamp0=not (flags & NC0), lowp="bf16")
            assert np.array_equal(multi.q[0].cpu().numpy(), ref["packed"])
