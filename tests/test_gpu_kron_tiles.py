"""GPU parity for fq_kron_tiles_kernel (csrc/fq_kron_tiles.hip, round 4): packed-only launches of the factor pairs whose token is not
four 32-column tiles wide — 80 x 112 (8960, Qwen2.5-1.5B ffn), 128 x 144 (18432, DeepSeek-V3 dense ffn), 144 x 192 (27648,
Qwen2.5-32B ffn) and every M of the same tile counts (function_utils.py:11-21 pairs).

It shares the fragment workspace and the quantiser helpers with the workgroup-per-token kernel; token staging (unpadded rows, a
bank rotation per row pitch, row indices clamped instead of zero rows), the half-empty last n'-tile, the streamed R fragments, the
meetings and the claims are its own. Every case is compared BIT FOR BIT with the workgroup-per-token kernel (a launch that also
asks for the transform takes that one) and with the oracle's quantiser on that transform.
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


def make(M, N, rows, seed, spike=True):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, M * N, generator=gen).half()
    if spike and rows:
        x[:, ::97] *= 20
    L = (torch.randn(M, M, generator=gen) / M ** 0.5).half()
    R = (torch.randn(N, N, generator=gen) / N ** 0.5).half()
    return x.cuda(), L.cuda(), R.cuda()


# (M, N): both K-step counts of every row-tile class, DMA tails (M * N / 8 % 64 != 0) and whole-instruction tokens
PAIRS = [(80, 112), (66, 112), (96, 112), (88, 112), (128, 144), (98, 144), (112, 144), (100, 144), (126, 144),
         (144, 192), (129, 192), (130, 192), (137, 192), (86, 128), (66, 128), (96, 128), (168, 176), (162, 176), (178, 176),
         (128, 148), (98, 148), (112, 148), (126, 148)]   # N % 16 != 0: rows of 296 bytes, a quarter-valid last half-tile


@pytest.mark.parametrize("M,N", PAIRS)
@pytest.mark.parametrize("rows", [1, 2, 3, 7, 100, 777])
def test_bit_equal_to_workgroup_kernel_and_oracle(ops, M, N, rows):
    x, L, R = make(M, N, rows, M * 7 + N + rows)
    sigs = [SIG, (0.9, 0.33), (1e-7, 1e-7)]     # magic-number route, clamp route, true-division route
    both = ops.kron_quant(x, L, R, sigs, T | P | R16)           # workgroup-per-token kernel (asks for the transform too)
    y16 = both.y.cpu().numpy().astype(np.float32)
    multi = ops.kron_quant(x, L, R, sigs, P | R16)              # this kernel, three clip sets in one launch
    for ci, sig in enumerate(sigs):
        one = ops.kron_quant(x, L, R, [sig], P | R16)           # this kernel, one clip set
        ref = O.quant_outputs(y16, sig[0], sig[1])
        for o, k in ((one, 0), (multi, ci)):
            assert torch.equal(o.q[k], both.q[ci]), (M, N, rows, sig)
            assert torch.equal(o.scale[k], both.scale[ci]), (M, N, rows, sig)
            assert np.array_equal(o.q[k].cpu().numpy(), ref["packed"]), (M, N, rows, sig)
            assert np.array_equal(o.scale[k].cpu().numpy(), ref["scale16"]), (M, N, rows, sig)


@pytest.mark.parametrize("M,N", [(80, 112), (128, 144), (144, 192), (128, 148), (168, 176)])
@pytest.mark.parametrize("flags", [P | NC0, P, P | R16 | NC0])
def test_flag_routes_bit_equal_to_workgroup_kernel(ops, M, N, flags):
    """Path A / path B rounding and the no-clamp statistics: same bits as the kernel that also returns the transform. The
    all-positive and all-negative tokens are where a padding lane's zero would show (the half-empty last tile at N = 112 / 144)."""
    x, L, R = make(M, N, 333, 5)
    x[3] = x[3].abs()
    a = ops.kron_quant(x, L, R, [SIG], flags)
    b = ops.kron_quant(x, L, R, [SIG], flags | T)
    assert torch.equal(a.q[0], b.q[0]) and torch.equal(a.scale[0], b.scale[0])
    # a transform whose values all have one sign: identity factors on a one-signed token
    Li, Ri = torch.eye(M, device="cuda").half(), torch.eye(N, device="cuda").half()
    for sgn in (1.0, -1.0):
        xp = (x.abs() + 0.5) * sgn
        a = ops.kron_quant(xp, Li, Ri, [SIG], flags)
        b = ops.kron_quant(xp, Li, Ri, [SIG], flags | T)
        assert torch.equal(a.q[0], b.q[0]) and torch.equal(a.scale[0], b.scale[0]), sgn


@pytest.mark.parametrize("M,N,rows", [(80, 112, 16384), (128, 144, 8192), (144, 192, 8192), (128, 148, 8192), (168, 176, 4096)])
def test_full_size_bit_equal_and_repeatable(ops, M, N, rows):
    """Full-size launches: every token equals the workgroup-per-token kernel's, and ten launches in a row give the same bytes
    (tokens are claimed dynamically: the schedule differs from launch to launch)."""
    x, L, R = make(M, N, rows, 11)
    ref = ops.kron_quant(x, L, R, [SIG], P | T | NC0)
    q0, s0 = ref.q[0].clone(), ref.scale[0].clone()
    del ref
    for _ in range(10):
        o = ops.kron_quant(x, L, R, [SIG], P | NC0)
        assert torch.equal(o.q[0], q0) and torch.equal(o.scale[0], s0)


@pytest.mark.parametrize("M,N", [(80, 112), (128, 144), (144, 192)])
def test_grouped_launch(ops, M, N):
    """Per-expert clip pairs (fq_kron_quant_grouped_f16) through this kernel: equal to one launch per group."""
    x, L, R = make(M, N, 700, 17)
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


def test_oracle_end_to_end_dyadic(ops):
    """Dyadic factors and tokens (every product and sum exact in fp16 / fp32): the whole launch equals the oracle's path-A
    transform + quantiser bit for bit, for each pair."""
    for M, N in ((80, 112), (128, 144), (144, 192), (128, 148), (168, 176), (86, 128)):
        gen = torch.Generator().manual_seed(M + N)
        x = (torch.randint(-8, 9, (64, M * N), generator=gen).float() / 8).half()
        L = torch.zeros(M, M)
        R = torch.zeros(N, N)
        L[torch.arange(M), torch.randperm(M, generator=gen)] = 1.0
        L[torch.arange(M), torch.randperm(M, generator=gen)] += 0.5
        R[torch.arange(N), torch.randperm(N, generator=gen)] = 1.0
        R[torch.arange(N), torch.randperm(N, generator=gen)] -= 0.25
        L, R = L.half(), R.half()
        o = ops.kron_quant(x.cuda(), L.cuda(), R.cuda(), [SIG], P | R16)
        ref = O.kron_quant(x.numpy(), L.numpy(), R.numpy(), SIG[0], SIG[1], round_y_f16=True)
        assert np.array_equal(o.q[0].cpu().numpy(), ref["packed"]), (M, N)
        assert np.array_equal(o.scale[0].cpu().numpy(), ref["scale16"]), (M, N)


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("M,N", [(80, 112), (128, 144), (112, 144), (144, 192), (130, 192), (112, 128), (86, 128), (128, 128), (100, 128), (128, 148),
                                 (110, 148), (168, 176)])
@pytest.mark.parametrize("rows", [1, 7, 300])
def test_bf16_packed_only_launches(ops, M, N, rows):
    """bf16 activations and factors (the DeepSeek-V3 flow runs under torch.set_default_dtype(bfloat16), main_dpskv3.py:395; 18432 =
    128 x 144 is its dense ffn): the same kernel with bf16 MFMA and bf16 rounding points — and, on bf16, the N = 128 pairs as well
    (fq_kron_trio.hip is fp16-only). Bit-equal, clip set by clip set and flag route by flag route, to the launch that also returns
    the transform (the workgroup-per-token kernel), and to the oracle's bf16 quantiser on that transform."""
    BF = torch.bfloat16
    gen = torch.Generator().manual_seed(M * 131 + N + rows)
    x = torch.randn(rows, M * N, generator=gen)
    x[:, ::97] *= 20
    x = x.to(BF).cuda()
    L = (torch.randn(M, M, generator=gen) / M ** 0.5).to(BF).cuda()
    R = (torch.randn(N, N, generator=gen) / N ** 0.5).to(BF).cuda()
    sigs = [SIG, (0.9, 0.33), (1e-7, 1e-7)]   # magic-number, clamp, true-division routes
    for flags in (P | NC0, P, P | R16 | NC0):
        both = ops.kron_quant(x, L, R, sigs, flags | T)
        multi = ops.kron_quant(x, L, R, sigs, flags)
        for ci, sig in enumerate(sigs):
            one = ops.kron_quant(x, L, R, [sig], flags)
            for o, k in ((one, 0), (multi, ci)):
                assert torch.equal(o.q[k], both.q[ci]), (M, N, rows, flags, sig)
                assert np.array_equal(_bits(o.scale[k]), _bits(both.scale[ci])), (M, N, rows, flags, sig)
        if flags & R16:   # the quantiser saw exactly the bf16 transform the other launch returned
            ref = O.quant_outputs(O.bf16_from_bits(_bits(both.y)), *sigs[0], round_y_f16=True, clamp0=not (flags & NC0), lowp="bf16")
            assert np.array_equal(multi.q[0].cpu().numpy(), ref["packed"])


def test_bf16_full_size_repeatable(ops):
    """DeepSeek-V3's dense ffn pair at full size on bf16: equal to the workgroup-per-token kernel on every token, and repeatable."""
    BF = torch.bfloat16
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(8192, 128 * 144, generator=gen).to(BF).cuda()
    L = (torch.randn(128, 128, generator=gen) / 128 ** 0.5).to(BF).cuda()
    R = (torch.randn(144, 144, generator=gen) / 12.0).to(BF).cuda()
    ref = ops.kron_quant(x, L, R, [SIG], P | T | NC0)
    q0, s0 = ref.q[0].clone(), ref.scale[0].clone()
    del ref
    for _ in range(5):
        o = ops.kron_quant(x, L, R, [SIG], P | NC0)
        assert torch.equal(o.q[0], q0) and np.array_equal(_bits(o.scale[0]), _bits(s0))
