"""GPU parity for the generic fused Kronecker kernel (every factor pair of SURVEY 8a other than 64x64).

Same bars as tests/test_gpu_kron64.py: dyadic fixtures bit-exact vs both reference paths; quantise/pack stage
bit-exact vs the oracle on the kernel's own transformed activation; transform within the north-star tolerance.
"""
import numpy as np
import pytest

from conftest import BOUND37, flip_ok
from tests.conftest import same_bits
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu
P, F, T, R16, NC0, Q16 = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20
SHAPES = ["64x128", "112x128", "128x224", "86x128", "64x112", "32x64", "56x64"]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def mismatch(a, b):
    return float(np.mean(np.asarray(a).reshape(-1) != np.asarray(b).reshape(-1)))


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("shape", ["64x128", "32x64"])
def test_exact_fixture_bit_exact_vs_both_reference_paths(ops, golden, shape):
    g = golden(f"exact_{shape}")
    x, L, Rm = dev(g["x"]), dev(g["L"]), dev(g["R"])
    rows = x.shape[0]
    for ci in range(2):
        sig = [(float(g[f"sig{ci}"][0]), float(g[f"sig{ci}"][1]))]
        o = ops.kron_quant(x, L, Rm, sig, P | NC0)
        assert np.array_equal(o.q[0].cpu().numpy(), g[f"b_packed{ci}"])
        assert np.array_equal(o.scale[0].cpu().numpy(), g[f"b_scale{ci}"])
        o = ops.kron_quant(x, L, Rm, sig, F | R16)
        assert same_bits(o.fq[0].cpu().numpy(), g[f"a_fq{ci}"].reshape(rows, -1))
        o = ops.kron_quant(x, L, Rm, sig, T)
        assert np.array_equal(o.y.cpu().numpy(), g[f"a_y{ci}"].reshape(rows, -1))


@pytest.mark.parametrize("shape", SHAPES)
def test_quant_stage_bit_exact_given_kernel_transform(ops, golden, shape):
    g = golden(f"kron_A_{shape}")
    x, L, Rm = dev(g["x"]), dev(g["L"]), dev(g["R"])
    sigs = [(0.9820137619972229, 0.9820137619972229), (0.9, 0.33)]
    o = ops.kron_quant(x, L, Rm, sigs, T | P | F | R16)
    y16 = o.y.cpu().numpy()
    for ci, (smax, smin) in enumerate(sigs):
        ref = O.quant_outputs(y16.astype(np.float32), smax, smin)
        assert np.array_equal(o.q[ci].cpu().numpy(), ref["packed"])
        assert np.array_equal(o.scale[ci].cpu().numpy(), ref["scale16"])
        assert same_bits(o.fq[ci].cpu().numpy(), ref["fq"])
    # fp16-arithmetic quantiser (lac = False path of quant_utils.py) on the same transform
    o = ops.kron_quant(x, L, Rm, [(1.0, 1.0)], T | F | R16 | Q16)
    ref = O.quant_outputs(o.y.cpu().numpy().astype(np.float32), 1.0, 1.0, quant_f16=True)
    assert same_bits(o.fq[0].cpu().numpy(), ref["fq"])


@pytest.mark.parametrize("shape", SHAPES)
def test_packed_only_kernels_quant_stage_bit_exact(ops, golden, shape):
    """Packed-only launches take their own kernels (one wave per token where a token fits a wave). Each of their
    quantiser routes -- magic-number rounding with / without clamp, true division -- and the multi-clip loop must
    reproduce the oracle on the transform the transform-only launch returns (same MFMA sequence, same bits)."""
    g = golden(f"kron_A_{shape}")
    x, L, Rm = dev(g["x"]), dev(g["L"]), dev(g["R"])
    y16 = ops.kron_quant(x, L, Rm, flags=T).y.cpu().numpy().astype(np.float32)
    sigs = [(0.9820137619972229, 0.9820137619972229), (0.9, 0.33), (1.0, 1.0), (1e-7, 1e-7)]
    for sig in sigs:
        o = ops.kron_quant(x, L, Rm, [sig], P | R16)
        ref = O.quant_outputs(y16, sig[0], sig[1])
        assert np.array_equal(o.q[0].cpu().numpy(), ref["packed"]), sig
        assert np.array_equal(o.scale[0].cpu().numpy(), ref["scale16"]), sig
    o = ops.kron_quant(x, L, Rm, sigs[:3], P | R16)
    for ci, sig in enumerate(sigs[:3]):
        ref = O.quant_outputs(y16, sig[0], sig[1])
        assert np.array_equal(o.q[ci].cpu().numpy(), ref["packed"]), ("multi", sig)
        assert np.array_equal(o.scale[ci].cpu().numpy(), ref["scale16"]), ("multi", sig)


@pytest.mark.parametrize("shape", SHAPES)
def test_transform_and_packed_vs_oracle_and_reference(ops, golden, shape):
    g = golden(f"kron_A_{shape}")
    x, L, Rm = dev(g["x"]), dev(g["L"]), dev(g["R"])
    rows = x.shape[0]
    y = ops.kron_quant(x, L, Rm, flags=T).y.cpu().numpy()
    y32 = O.kron_transform(g["x"], g["L"], g["R"]).reshape(rows, -1)
    assert mismatch(y, y32.astype(np.float16)) <= 1e-2
    den = np.abs(y32).max(axis=1, keepdims=True)
    assert np.max(np.abs(y.astype(np.float32) - y32) / den) <= 1e-3
    assert mismatch(y, g["a16_lac0_y"]) <= 2e-2                      # reference path A (torch CPU BLAS order)
    for ci in range(2):
        s = (float(g["sig"][ci][0]), float(g["sig"][ci][1]))
        o = ops.kron_quant(x, L, Rm, [s], P | R16)
        q = O.unpack_i4(o.q[0].cpu().numpy())
        ref = O.kron_quant(g["x"], g["L"], g["R"], s[0], s[1], round_y_f16=True)
        assert mismatch(q, ref["q"]) <= 1e-3 and np.max(np.abs(q - ref["q"].astype(np.int32))) <= 1
        qa = g[f"a16_lac{ci}_q"].reshape(rows, -1).astype(np.int32)
        assert mismatch(q, qa) <= 1e-3 and np.max(np.abs(q - qa)) <= 1
        sg = o.scale[0].cpu().numpy().astype(np.float32)
        assert np.max(np.abs(sg - ref["scale"]) / ref["scale"]) <= 1e-3


@pytest.mark.parametrize("shape", ["64x128", "32x64", "112x128"])
def test_vs_reference_path_b_packed(ops, golden, shape):
    g = golden(f"kron_B_{shape}")
    x, L, Rm = dev(g["x"]), dev(g["L"]), dev(g["R"])
    for ci in range(3):
        s = (float(g["sig"][ci][0]), float(g["sig"][ci][1]))
        o = ops.kron_quant(x, L, Rm, [s], P | NC0)
        q, qb = O.unpack_i4(o.q[0].cpu().numpy()), O.unpack_i4(g[f"b_packed{ci}"])
        assert mismatch(q, qb) <= 1e-3 and np.max(np.abs(q - qb)) <= 1
        sb = g[f"b_scale{ci}"].astype(np.float32)
        assert np.max(np.abs(o.scale[0].cpu().numpy().astype(np.float32) - sb) / sb) <= 1e-3


@pytest.mark.parametrize("M,N,rows", [(64, 128, 0), (64, 128, 1), (112, 128, 3), (128, 224, 5), (32, 64, 1025),
                                      (64, 80, 7), (128, 128, 9), (96, 96, 4), (128, 256, 3), (128, 144, 5), (80, 112, 3),
                                      (32, 48, 9), (100, 144, 2)])
def test_ragged_rows_and_other_factor_pairs(ops, M, N, rows):
    gen = torch.Generator().manual_seed(M * 1000 + N + rows)
    x = torch.randn(rows, M * N, generator=gen).half()
    L = (torch.randn(M, M, generator=gen) / (M ** 0.5)).half()
    Rm = (torch.randn(N, N, generator=gen) / (N ** 0.5)).half()
    o = ops.kron_quant(x.cuda(), L.cuda(), Rm.cuda(), [(0.97, 0.9)], T | P | R16)
    assert o.q[0].shape == (rows, M * N // 2)
    if rows:
        n = min(rows, 4)
        ref = O.quant_outputs(o.y[-n:].cpu().numpy().astype(np.float32), 0.97, 0.9)
        assert np.array_equal(o.q[0][-n:].cpu().numpy(), ref["packed"])
        y32 = O.kron_transform(x[-n:].numpy(), L.numpy(), Rm.numpy()).reshape(n, -1)
        assert mismatch(o.y[-n:].cpu().numpy(), y32.astype(np.float16)) <= 1e-2


def test_single_signed_token_no_clamp_masks_padding(ops):
    """NO_CLAMP0 statistics must ignore the zero padding of non-multiple-of-32 factors (M = 86, 112)."""
    M, N = 86, 128
    x = torch.zeros(2, M * N).half()
    x[:, :] = 1.0                                             # all-positive token
    L, Rm = torch.eye(M).half(), torch.eye(N).half()          # identity transform: Y == X > 0 everywhere
    o = ops.kron_quant(x.cuda(), L.cuda(), Rm.cuda(), [(1.0, 0.5)], P | NC0)
    ref = O.kron_quant(x.numpy(), L.numpy(), Rm.numpy(), 1.0, 0.5, clamp0=False)
    assert np.array_equal(o.q[0].cpu().numpy(), ref["packed"])
    assert np.array_equal(o.scale[0].cpu().numpy(), ref["scale16"])


def test_unsupported_shape_raises(ops):
    from flatquant_amd._lib import FqError
    x = torch.randn(2, 60 * 63).half().cuda()
    with pytest.raises(FqError):                               # odd element count: nothing to pack two per byte
        ops.kron_quant(x, torch.eye(60).half().cuda(), torch.eye(63).half().cuda())
    with pytest.raises(FqError):                               # beyond the general kernel (M * N > 32768)
        ops.kron_quant(torch.zeros(1, 200 * 200).half().cuda(), torch.eye(200).half().cuda(), torch.eye(200).half().cuda())


@pytest.mark.parametrize("M,N", [(128, 148), (144, 192), (168, 176), (60, 62), (3, 6), (130, 16), (20, 12), (64, 80), (96, 96),
                                 (256, 128), (192, 170), (2, 2), (129, 254), (40, 100)])
def test_general_kernel_any_pair(ops, M, N):
    """Factor pairs the specialised kernels do not take (Qwen2.5 ffn widths 18944 = 128 x 148, 27648 = 144 x 192,
    29568 = 168 x 176; N % 16 != 0, N % 4 != 0, M > 128, tokens of 36 bytes) run the general MFMA kernel
    (csrc/fq_kron_general.hip): same bars — transform within 1e-3 of the row maximum, every quantiser output bit-exact on
    the kernel's own transform, packed result vs the oracle pipeline."""
    gen = torch.Generator().manual_seed(M * 1000 + N)
    rows = 5
    x = torch.randn(rows, M * N, generator=gen).half()
    x[1] = 0
    x[2] = x[2].abs()
    L = (torch.randn(M, M, generator=gen) / (M ** 0.5)).half()
    Rm = (torch.randn(N, N, generator=gen) / (N ** 0.5)).half()
    sigs = [(0.97, 0.9), (1.0, 1.0)]
    o = ops.kron_quant(x.cuda(), L.cuda(), Rm.cuda(), sigs, T | P | F | R16)
    y16 = o.y.cpu().numpy()
    y32 = O.kron_transform(x.numpy(), L.numpy(), Rm.numpy()).reshape(rows, -1)
    den = np.abs(y32).max(axis=1, keepdims=True) + 1e-30
    assert np.max(np.abs(y16.astype(np.float32) - y32) / den) <= 1e-3
    for ci, (smax, smin) in enumerate(sigs):
        ref = O.quant_outputs(y16.astype(np.float32), smax, smin)
        assert np.array_equal(o.q[ci].cpu().numpy(), ref["packed"])
        assert np.array_equal(o.scale[ci].cpu().numpy(), ref["scale16"])
        assert same_bits(o.fq[ci].cpu().numpy(), ref["fq"])
    o2 = ops.kron_quant(x.cuda(), L.cuda(), Rm.cuda(), [sigs[0]], P | NC0)
    ref = O.kron_quant(x.numpy(), L.numpy(), Rm.numpy(), sigs[0][0], sigs[0][1], clamp0=False)
    q = O.unpack_i4(o2.q[0].cpu().numpy())
    assert flip_ok(q, ref["q"].astype(np.int32), f"kron generic vs oracle {M}x{N}", BOUND37)
    d = torch.rand(M * N, generator=gen).half() + 0.5
    od = ops.kron_quant(x.cuda(), L.cuda(), Rm.cuda(), flags=T, diag=d.cuda()).y.cpu().numpy()
    yd = O.kron_transform((x * d).numpy(), L.numpy(), Rm.numpy()).reshape(rows, -1)
    assert np.max(np.abs(od.astype(np.float32) - yd) / (np.abs(yd).max(axis=1, keepdims=True) + 1e-30)) <= 1e-3


def test_prepared_workspace_reuse_and_invalidation(ops):
    """The fragment re-pack of (left, right) is skipped when the same two buffers come back (FQ_WS_PREPARED); an
    in-place update of a matrix must invalidate that, and fq_kron_prepare_f16 + the flag must equal a plain call."""
    import ctypes
    from flatquant_amd._lib import FQ_WS_PREPARED, check, lib
    gen = torch.Generator().manual_seed(5)
    M, N, rows = 64, 128, 33
    x = torch.randn(rows, M * N, generator=gen).half().cuda()
    L = (torch.randn(M, M, generator=gen) / 8).half().cuda()
    R = (torch.randn(N, N, generator=gen) / 11).half().cuda()
    sig = [(0.95, 0.9)]
    a = ops.kron_quant(x, L, R, sig, P)
    b = ops.kron_quant(x, L, R, sig, P)            # second call: workspace reused
    assert torch.equal(a.q[0], b.q[0]) and torch.equal(a.scale[0], b.scale[0])
    L.mul_(-1.0)                                   # in-place: same address, new version
    c = ops.kron_quant(x, L, R, sig, P)
    ref = ops.kron_quant(x, L.clone(), R.clone(), sig, P)
    assert torch.equal(c.q[0], ref.q[0]) and torch.equal(c.scale[0], ref.scale[0])
    assert not torch.equal(c.q[0], a.q[0])
    # the C ABI directly: prepare once, then FQ_WS_PREPARED with garbage-proofing (matrix pointers still passed)
    nbytes = int(lib.fq_kron_workspace_bytes(M, N))
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.fq_kron_prepare_f16(L.data_ptr(), R.data_ptr(), M, N, ws.data_ptr(), nbytes, st))
    q = torch.empty(rows, M * N // 2, dtype=torch.uint8, device="cuda")
    s = torch.empty(rows, dtype=torch.float16, device="cuda")
    qa, sa, none4 = (ctypes.c_void_p * 4)(), (ctypes.c_void_p * 4)(), (ctypes.c_void_p * 4)()
    qa[0], sa[0] = q.data_ptr(), s.data_ptr()
    smax, smin = (ctypes.c_float * 4)(sig[0][0]), (ctypes.c_float * 4)(sig[0][1])
    check(lib.fq_kron_quant_f16(x.data_ptr(), L.data_ptr(), R.data_ptr(), None, rows, M, N, smax, smin, 1,
                                P | FQ_WS_PREPARED, qa, sa, none4, None, ws.data_ptr(), nbytes, st))
    assert torch.equal(q, c.q[0]) and torch.equal(s, c.scale[0])
    assert lib.fq_kron_prepare_f16(L.data_ptr(), R.data_ptr(), M, N, ws.data_ptr(), 16, st) < 0   # too small


def test_general_kernel_grouped_and_post_scale(ops):
    """Grouped (per-expert clip pairs) launches and the post-scale of fq_kron_quant_ex_f16 on a pair only the general kernel
    takes (128 x 148)."""
    M, N, rows = 128, 148, 50
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(rows, M * N, generator=gen).half().cuda()
    L = (torch.randn(M, M, generator=gen) / M ** 0.5).half().cuda()
    Rm = (torch.randn(N, N, generator=gen) / N ** 0.5).half().cuda()
    offs = torch.tensor([0, 10, 10, 50], dtype=torch.int64, device="cuda")
    smax = torch.tensor([0.9, 0.5, 0.7], device="cuda")
    smin = torch.tensor([0.8, 0.5, 1.0], device="cuda")
    o = ops.kron_quant_grouped(x, L, Rm, offs, smax, smin, P | NC0)
    for g in (0, 2):
        a, b = int(offs[g]), int(offs[g + 1])
        one = ops.kron_quant(x[a:b].contiguous(), L, Rm, [(float(smax[g]), float(smin[g]))], P | NC0)
        assert torch.equal(o.q[0][a:b], one.q[0]) and torch.equal(o.scale[0][a:b], one.scale[0])
    ps = 0.37
    y0 = ops.kron_quant(x, L, Rm, flags=T).y.float().cpu().numpy()
    ex = ops.kron_quant_ex(x, L, Rm, ps, [(0.97, 0.9)], T | P | R16)
    y = ex.y.cpu().numpy().astype(np.float32)
    assert np.max(np.abs(y - y0 * ps) / (np.abs(y0 * ps).max(axis=1, keepdims=True))) <= 2e-3
    ref = O.quant_outputs(y, 0.97, 0.9)
    assert np.array_equal(ex.q[0].cpu().numpy(), ref["packed"]) and np.array_equal(ex.scale[0].cpu().numpy(), ref["scale16"])


@pytest.mark.parametrize("rows", [1, 2, 11, 12, 13, 500, 3073])
def test_wave_kernel_64x80_packed_only(ops, rows):
    """64 x 80 (5120 = Qwen2.5-14B/32B hidden; kernel_benchmark.py:234-246 lists it): the packed-only launch runs the
    wave-per-token kernel with THREE column tiles (80 = 2.5 x 32: a lane's run is 24 bytes, the h = 1 half holds 32 valid
    n' of 48) — bit-equal to the launch that also returns the transform, and to the oracle's quantiser on it, on the
    three quantiser routes, with and without the clamp of the extrema through zero."""
    gen = torch.Generator().manual_seed(80 + rows)
    x = torch.randn(rows, 64 * 80, generator=gen).half()
    x[:, ::41] *= 25
    if rows > 1:
        x[1] = x[1].abs() + 1                                   # single-signed token: NO_CLAMP0 must not see the padding columns
    L = (torch.randn(64, 64, generator=gen) / 8).half().cuda()
    Rm = (torch.randn(80, 80, generator=gen) / 80 ** 0.5).half().cuda()
    if rows > 1:
        L, Rm = L.abs(), Rm.abs()                               # keeps that token's transform positive
    sigs = [(0.982, 0.982), (0.9, 0.33), (1e-7, 1e-7)]
    for fl, clamp0 in ((R16, True), (R16 | NC0, False)):
        both = ops.kron_quant(x.cuda(), L, Rm, sigs, T | P | fl)
        only = ops.kron_quant(x.cuda(), L, Rm, sigs, P | fl)
        y = both.y.cpu().numpy().astype(np.float32)
        for ci, s in enumerate(sigs):
            ref = O.quant_outputs(y, s[0], s[1], clamp0=clamp0)
            assert np.array_equal(only.q[ci].cpu().numpy(), ref["packed"]), (rows, fl, s)
            assert np.array_equal(only.scale[ci].cpu().numpy(), ref["scale16"]), (rows, fl, s)
            assert torch.equal(only.q[ci], both.q[ci]) and torch.equal(only.scale[ci], both.scale[ci])
    again = ops.kron_quant(x.cuda(), L, Rm, sigs, P | R16 | NC0)  # a second launch: same bytes (prefetch / counted waits)
    assert all(torch.equal(a, b) for a, b in zip(again.q, only.q))


@pytest.mark.parametrize("M", [104, 120, 128])
@pytest.mark.parametrize("rows", [1, 3, 300, 1025])
def test_workgroup_kernel_n148_packed_only(ops, M, rows):
    """N = 148 (18944 = 128 x 148, Qwen2.5-7B ffn; N % 16 != 0): the packed-only launch runs the workgroup-per-token
    kernel (8-byte staging units, a last 16-column run cut to 4 columns, 74-byte packed rows) — bit-equal to the launch
    that also returns the transform (general kernel) and to the oracle's quantiser on that transform, on the three
    quantiser routes, with and without the clamp of the extrema through zero (the cut run must not leak its padding)."""
    gen = torch.Generator().manual_seed(148 + M + rows)
    x = torch.randn(rows, M * 148, generator=gen).half()
    x[:, ::43] *= 25
    L = (torch.randn(M, M, generator=gen) / M ** 0.5).half().cuda()
    Rm = (torch.randn(148, 148, generator=gen) / 148 ** 0.5).half().cuda()
    if rows > 1:
        x[1] = x[1].abs() + 1                                   # single-signed token
        L, Rm = L.abs(), Rm.abs()
    sigs = [(0.982, 0.982), (0.9, 0.33), (1e-7, 1e-7)]
    for fl, clamp0 in ((R16, True), (R16 | NC0, False)):
        both = ops.kron_quant(x.cuda(), L, Rm, sigs, T | P | fl)
        only = ops.kron_quant(x.cuda(), L, Rm, sigs, P | fl)
        y = both.y.cpu().numpy().astype(np.float32)
        for ci, s in enumerate(sigs):
            ref = O.quant_outputs(y, s[0], s[1], clamp0=clamp0)
            assert np.array_equal(only.q[ci].cpu().numpy(), ref["packed"]), (M, rows, fl, s)
            assert np.array_equal(only.scale[ci].cpu().numpy(), ref["scale16"]), (M, rows, fl, s)
    again = ops.kron_quant(x.cuda(), L, Rm, sigs, P | R16 | NC0)
    assert all(torch.equal(a, b) for a, b in zip(again.q, only.q))


@pytest.mark.parametrize("M,N", [(64, 128), (64, 112), (64, 80), (56, 64), (32, 64), (37, 64)])
@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_wave_kernel_fake_quant_and_transform_outputs(ops, M, N, dt):
    """Round 4: fake-quant-only and transform-only launches of the wave-per-token pairs run fq_kron_wave_kernel<..., OS> (they ran the
    workgroup-per-token kernel). Bit-equal to the launch that asks for BOTH outputs (which still takes that kernel), clip route by clip
    route, with and without the rounding of Y, fp16 and bf16; and the fake-quant values are the oracle's on the returned transform."""
    td = torch.float16 if dt == "f16" else torch.bfloat16
    gen = torch.Generator().manual_seed(M * 17 + N)
    for rows in (1, 7, 300):
        x = torch.randn(rows, M * N, generator=gen)
        x[:, ::97] *= 20
        x = x.to(td).cuda()
        L = (torch.randn(M, M, generator=gen) / M ** 0.5).to(td).cuda()
        R = (torch.randn(N, N, generator=gen) / N ** 0.5).to(td).cuda()
        for sig in ((0.9820137619972229, 0.9820137619972229), (0.9, 0.33), (1e-7, 1e-7)):
            for fl in (F | R16, F | R16 | NC0, F):
                both = ops.kron_quant(x, L, R, [sig], fl | T)
                fq = ops.kron_quant(x, L, R, [sig], fl)
                assert torch.equal(fq.fq[0].view(torch.int16), both.fq[0].view(torch.int16)), (M, N, rows, sig, fl)
                if fl & R16:
                    y = ops.kron_quant(x, L, R, flags=T | R16)
                    assert torch.equal(y.y.view(torch.int16), both.y.view(torch.int16)), (M, N, rows)
        if dt == "f16":
            sg = (0.9820137619972229, 0.9820137619972229)
            b2 = ops.kron_quant(x, L, R, [sg], F | R16 | T)
            ref = O.quant_outputs(b2.y.cpu().numpy().astype(np.float32), sg[0], sg[1], round_y_f16=False, clamp0=True)
            assert same_bits(ops.kron_quant(x, L, R, [sg], F | R16).fq[0].cpu().numpy(), ref["fq"])
