"""The INT4 GEMM / Linear4bit on the FP6 matrix path (fq_int4_to_bf6 + fq_bf6_gemm_i32 / fq_bf6_linear_f16): integer
work carried by exactly representable BF6 values and an fp32 accumulator that stays an integer — the bar is bit-exact
against the integer oracle and against the int8-path kernel."""
import numpy as np
import pytest
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def rand_packed(gen, rows, K, lo=-8, hi=8):
    q = torch.randint(lo, hi, (rows, K), generator=gen, dtype=torch.int32).numpy()
    return O.pack_i4(q), q


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 256, 4096), (300, 272, 256), (1, 16, 128), (257, 4096, 512),
                                   (1000, 1024, 1024), (33, 48, 384), (64, 256, 14336), (31, 32, 128), (513, 528, 640)])
def test_bf6_gemm_bit_exact(ops, M, N, K):
    gen = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    xp, xq = rand_packed(gen, M, K)
    wp, wq = rand_packed(gen, N, K)
    xb = ops.int4_to_bf6(torch.from_numpy(xp).cuda())
    wb = ops.int4_to_bf6(torch.from_numpy(wp).cuda(), weights=True)
    c = ops.bf6_matmul(xb, wb, M, N, K).cpu().numpy()
    ref = xq.astype(np.int64) @ wq.astype(np.int64).T
    assert np.array_equal(c, ref.astype(np.int32))


def test_every_value_pair_and_extremes(ops):
    """all 256 (x, w) nibble pairs isolated one per output, then all -8 x -8 at K = 14336 (917504 per output < 2^24)."""
    vals = np.arange(-8, 8)
    M = N = 16
    K = 128
    xq = np.zeros((M, K), dtype=np.int32)
    wq = np.zeros((N, K), dtype=np.int32)
    xq[:, 5] = vals
    wq[:, 5] = vals
    xb = ops.int4_to_bf6(torch.from_numpy(O.pack_i4(xq)).cuda())
    wb = ops.int4_to_bf6(torch.from_numpy(O.pack_i4(wq)).cuda(), weights=True)
    c = ops.bf6_matmul(xb, wb, M, N, K).cpu().numpy()
    assert np.array_equal(c, np.outer(vals, vals))
    M, N, K = 40, 32, 14336
    xb = ops.int4_to_bf6(torch.from_numpy(O.pack_i4(np.full((M, K), -8, dtype=np.int32))).cuda())
    wb = ops.int4_to_bf6(torch.from_numpy(O.pack_i4(np.full((N, K), -8, dtype=np.int32))).cuda(), weights=True)
    assert np.all(ops.bf6_matmul(xb, wb, M, N, K).cpu().numpy() == 64 * K)
    wb = ops.int4_to_bf6(torch.from_numpy(O.pack_i4(np.full((N, K), 7, dtype=np.int32))).cuda(), weights=True)
    assert np.all(ops.bf6_matmul(xb, wb, M, N, K).cpu().numpy() == -56 * K)


@pytest.mark.parametrize("M,N,K,bias", [(512, 256, 4096, False), (300, 272, 256, True), (129, 1024, 1024, True)])
def test_bf6_linear_equals_int8_path_and_oracle(ops, M, N, K, bias):
    gen = torch.Generator().manual_seed(M + N + K)
    xp, _ = rand_packed(gen, M, K)
    wp, _ = rand_packed(gen, N, K)
    sx = (torch.rand(M, generator=gen) * 0.05 + 0.001).half()
    sw = (torch.rand(N, generator=gen) * 0.02 + 0.0005).half()
    b = torch.randn(N, generator=gen).half() if bias else None
    x, w = torch.from_numpy(xp).cuda(), torch.from_numpy(wp).cuda()
    bb = None if b is None else b.cuda()
    y = ops.bf6_linear(ops.int4_to_bf6(x), sx.cuda(), ops.int4_to_bf6(w, weights=True), sw.cuda(), bb, M, N, K)
    y8 = ops.int4_linear(x, sx.cuda(), w, sw.cuda(), bb)
    assert torch.equal(y, y8)
    ref = O.linear4bit(xp, sx.numpy(), wp, sw.numpy(), None if b is None else b.numpy())
    assert np.array_equal(y.cpu().numpy().view(np.uint16), ref.view(np.uint16))


@pytest.mark.parametrize("M,N,K", [(4096 + 37, 4096 + 16, 256), (8192, 2048, 128), (2048 + 1, 8192 + 32, 384)])
def test_persistent_workgroups_many_tiles_bit_exact(ops, M, N, K):
    """>= 192 tiles of 256 x 256: the 8-wave kernel with persistent workgroups (more tiles than CUs: a workgroup walks two or three,
    the next tile's stages requested in front of the epilogue), edge tiles in both dimensions, K loops of 1 - 3 stages."""
    assert ((M + 255) // 256) * ((N + 255) // 256) >= 192
    gen = torch.Generator().manual_seed(M + 3 * N + K)
    xp, xq = rand_packed(gen, M, K)
    wp, wq = rand_packed(gen, N, K)
    c = ops.bf6_matmul(ops.int4_to_bf6(torch.from_numpy(xp).cuda()), ops.int4_to_bf6(torch.from_numpy(wp).cuda(), weights=True), M, N, K)
    ref = (xq.astype(np.float64) @ wq.astype(np.float64).T).astype(np.int32)      # |sum| <= 64 K: exact in fp64
    assert np.array_equal(c.cpu().numpy(), ref)


@pytest.mark.parametrize("M,N,K,bias", [(4096, 3072, 11776, True), (8192 + 5, 4096, 384, True), (16384, 4096, 256, False),
                                        (1536 + 9, 4096, 11776, False)])
def test_linear_many_tiles_and_the_clamp_variant(ops, M, N, K, bias):
    """The fused sym_dequant epilogue behind the persistent K loop (interior tiles leave their stores in flight: counted vmcnt) and
    behind the 128-token tiles; K > 10176 takes the epilogue with the +-65176 clamp, and full-scale rows make it bind on both sides
    (-8 x -8 x 11776 = 753664, -8 x 7 x 11776 = -659456; |q| / 10 > 65176). Same bits as the int8-path kernel everywhere, and as
    the oracle on a sample of token rows."""
    gen = torch.Generator().manual_seed(M + N + K)
    xq = torch.randint(-8, 8, (M, K), generator=gen, dtype=torch.int32).numpy()
    wq = torch.randint(-8, 8, (N, K), generator=gen, dtype=torch.int32).numpy()
    if K > 10176:
        xq[::3] = -8
        wq[::4] = -8
        wq[1::4] = 7
    xp, wp = O.pack_i4(xq), O.pack_i4(wq)
    sx = (torch.rand(M, generator=gen) * 0.05 + 0.001).half()
    sw = (torch.rand(N, generator=gen) * 0.02 + 0.0005).half()
    b = torch.randn(N, generator=gen).half() if bias else None
    x, w = torch.from_numpy(xp).cuda(), torch.from_numpy(wp).cuda()
    bb = None if b is None else b.cuda()
    y = ops.bf6_linear(ops.int4_to_bf6(x), sx.cuda(), ops.int4_to_bf6(w, weights=True), sw.cuda(), bb, M, N, K)
    assert torch.equal(y.view(torch.int16), ops.int4_linear(x, sx.cuda(), w, sw.cuda(), bb).view(torch.int16))
    rows = np.unique(np.concatenate([np.arange(0, M, max(1, M // 24)), [M - 1]]))
    ref = O.linear4bit(xp[rows], sx.numpy()[rows], wp, sw.numpy(), None if b is None else b.numpy())
    assert np.array_equal(y.cpu().numpy()[rows].view(np.uint16), ref.view(np.uint16))


def test_module_uses_the_fp6_path_transparently(ops, monkeypatch):
    import flatquant_amd.deploy as deploy
    gen = torch.Generator().manual_seed(3)
    lin = deploy.nn.Linear4bit(512, 384, bias=True).cuda()
    lin.fp6_image, lin.fp6_min_out_features = True, 0      # policy attributes of the instance (no environment variable: round 4)
    wp, _ = rand_packed(gen, 384, 512)
    lin.weight.copy_(torch.from_numpy(wp))
    lin.weight_scales.copy_((torch.rand(384, 1, generator=gen) * 0.02 + 0.001))
    lin.bias.copy_(torch.randn(384, generator=gen).half())
    xp, _ = rand_packed(gen, 2 * 9, 512)
    p = deploy.PackedQuantizedTensor(torch.from_numpy(xp).cuda().reshape(2, 9, 256),
                                     (torch.rand(2, 1, 9, generator=gen) * 0.05 + 0.001).half().cuda())
    y = lin(p)
    assert lin._weight_image() is not None      # the FP6 path really ran
    ref = ops.int4_linear(p.quantized_x.reshape(-1, 256), p.scales_x.reshape(-1), lin.weight,
                          lin.weight_scales.reshape(-1).half(), lin.bias.half()).view(2, 9, 384)
    assert torch.equal(y, ref)
    lin.weight.copy_(torch.from_numpy(rand_packed(gen, 384, 512)[0]))      # in-place update: the cached image must follow
    ref2 = ops.int4_linear(p.quantized_x.reshape(-1, 256), p.scales_x.reshape(-1), lin.weight,
                           lin.weight_scales.reshape(-1).half(), lin.bias.half()).view(2, 9, 384)
    assert torch.equal(lin(p), ref2) and not torch.equal(ref, ref2)


def test_prefill_sized_calls_take_the_fp6_path_with_a_transient_image(ops, monkeypatch):
    """Default policy (no image kept): >= fp6_transient_rows tokens convert the weights for the call (inside the one library call
    fq_int4_linear_fp6_f16); same bits as the int8 path, nothing cached on the module."""
    import flatquant_amd.deploy as deploy
    gen = torch.Generator().manual_seed(5)
    lin = deploy.nn.Linear4bit(256, 272).cuda()
    lin.fp6_min_out_features = 0                            # (a 272-wide layer: below the default width gate)
    lin.fp6_image = False                                   # (round 5: the kept image is the default; this is the transient route)
    lin.weight.copy_(torch.from_numpy(rand_packed(gen, 272, 256)[0]))
    lin.weight_scales.copy_((torch.rand(272, 1, generator=gen) * 0.02 + 0.001))
    rows = lin.fp6_transient_rows + 3
    xp, _ = rand_packed(gen, rows, 256)
    p = deploy.PackedQuantizedTensor(torch.from_numpy(xp).cuda().reshape(1, rows, 128),
                                     (torch.rand(1, 1, rows, generator=gen) * 0.05 + 0.001).half().cuda())
    calls = []
    real = ops.int4_linear_fp6
    monkeypatch.setattr(ops, "int4_linear_fp6", lambda *a: (calls.append(a[3] is None), real(*a))[1])
    y = lin(p)
    assert calls == [True] and lin.image_bytes() == 0                  # the weights were converted for the call and not kept
    ref = ops.int4_linear(p.quantized_x.reshape(-1, 128), p.scales_x.reshape(-1), lin.weight, lin.weight_scales.reshape(-1).half(),
                          None).view(1, rows, 272)
    assert torch.equal(y, ref)
    calls.clear()
    lin.fp6_transient_rows = 0                                          # policy off: the int8 matrix path
    assert torch.equal(lin(p), ref) and not calls
    lin.fp6_image = True                                                # a kept image: the same entry point, no weight conversion
    lin._weight_image = lambda: ops.int4_to_bf6(lin.weight, weights=True)
    assert torch.equal(lin(p), ref) and calls == [False]


@pytest.mark.parametrize("M,N,K,bias,keep", [(300, 272, 256, True, False), (300, 272, 256, False, True), (2500, 4096, 512, True, True)])
def test_one_call_entry_point(ops, M, N, K, bias, keep):
    """fq_int4_linear_fp6_f16 (conversions + GEMM in one library call, with a kept or a transient weight image) == fq_int4_linear_f16."""
    gen = torch.Generator().manual_seed(M + N + K + keep)
    x, w = torch.from_numpy(rand_packed(gen, M, K)[0]).cuda(), torch.from_numpy(rand_packed(gen, N, K)[0]).cuda()
    sx = (torch.rand(M, generator=gen) * 0.05 + 0.001).half().cuda()
    sw = (torch.rand(N, generator=gen) * 0.02 + 0.0005).half().cuda()
    b = torch.randn(N, generator=gen).half().cuda() if bias else None
    img = ops.int4_to_bf6(w, weights=True) if keep else None
    assert torch.equal(ops.int4_linear_fp6(x, sx, w, img, sw, b).view(torch.int16), ops.int4_linear(x, sx, w, sw, b).view(torch.int16))
    with pytest.raises(Exception):
        ops.int4_linear_fp6(x[:, :32].contiguous(), sx, w[:, :32].contiguous(), None, sw, b)      # K = 64


def test_errors(ops):
    with pytest.raises(Exception):
        ops.int4_to_bf6(torch.zeros(4, 16, dtype=torch.uint8, device="cuda"))          # K = 32
    xb = ops.int4_to_bf6(torch.zeros(4, 32, dtype=torch.uint8, device="cuda"))
    wb = ops.int4_to_bf6(torch.zeros(16, 32, dtype=torch.uint8, device="cuda"), weights=True)
    with pytest.raises(Exception):
        ops.bf6_matmul(xb, wb, 4, 16, 64)                                               # K % 128 != 0


@pytest.mark.parametrize("M,Ns,K", [(300, (272, 256, 16), 256), (2048, (4096, 1024, 1024), 512), (129, (256,), 128), (2500, (1024, 1040), 384),
                                    (4096 + 3, (4096, 4096, 4096, 512), 256), (16384, (2048, 2048), 128)])
def test_multi_problem_launch_bit_identical_to_single_launches(ops, M, Ns, K):
    """fq_int4_linear_fp6_multi_f16 (round 4): up to four problems with common M and K — each with its own activations, weights, scales,
    bias, with a kept or a transient weight image — as one GEMM launch == one fq_int4_linear_fp6_f16 call per problem, bit for bit
    (tile boundaries between problems that are not multiples of the 256-wide tile, 128- and 256-token tiles, shared activations)."""
    gen = torch.Generator().manual_seed(M + sum(Ns) + K)
    problems, want = [], []
    x_shared = torch.from_numpy(rand_packed(gen, M, K)[0]).cuda()
    for p, N in enumerate(Ns):
        x = x_shared if p == 2 else torch.from_numpy(rand_packed(gen, M, K)[0]).cuda()     # (problems 0.. own x; problem 2 shares one)
        if p == 0:
            x_shared = x
        if p == 2:
            x = problems[0][0]
        w = torch.from_numpy(rand_packed(gen, N, K)[0]).cuda()
        sx = (torch.rand(M, generator=gen) * 0.05 + 0.001).half().cuda()
        sw = (torch.rand(N, generator=gen) * 0.02 + 0.0005).half().cuda()
        b = torch.randn(N, generator=gen).half().cuda() if p % 2 == 0 else None
        img = ops.int4_to_bf6(w, weights=True) if p % 2 == 1 else None
        problems.append((x, sx, w, img, sw, b))
        want.append(ops.int4_linear_fp6(x, sx, w, img, sw, b))
    got = ops.int4_linear_fp6_multi(problems)
    assert len(got) == len(Ns)
    for p in range(len(Ns)):
        assert got[p].shape == (M, Ns[p]) and torch.equal(got[p].view(torch.int16), want[p].view(torch.int16)), p
    with pytest.raises(Exception):
        ops.int4_linear_fp6_multi(problems * 5)      # more than four


def test_linear4bit_multi_module_entry(ops):
    """deploy.nn.linear.linear4bit_multi: q / k / v modules on their own PackedQuantizedTensors in one launch == m(x) each; decode-sized
    inputs and modules the FP6 route does not take fall back to the modules' own forward."""
    from flatquant_amd.deploy import PackedQuantizedTensor
    from flatquant_amd.deploy.nn.linear import Linear4bit, linear4bit_multi
    gen = torch.Generator().manual_seed(9)
    K = 512
    mods = []
    for N in (2048, 1024, 4096):      # (a narrow k / v projection rides along with the wide ones)
        m = Linear4bit(K, N, bias=N == 4096).cuda()
        m.weight.copy_(torch.from_numpy(rand_packed(gen, N, K)[0]))
        m.weight_scales.copy_((torch.rand(N, 1, generator=gen) * 0.02 + 0.0005))
        if m.bias is not None:
            m.bias.copy_(torch.randn(N, generator=gen).half())
        mods.append(m)
    mods[1].fp6_image = True
    for rows in (2048, 64):
        xs = [PackedQuantizedTensor(torch.from_numpy(rand_packed(gen, rows, K)[0]).cuda().reshape(2, rows // 2, K // 2),
                                    (torch.rand(2, rows // 2, 1, generator=gen) * 0.05 + 0.001).half().cuda()) for _ in mods]
        ys = linear4bit_multi(mods, xs)
        for m, x, y in zip(mods, xs, ys):
            ref = m(x)
            assert y.shape == ref.shape and torch.equal(y.view(torch.int16), ref.view(torch.int16)), (rows, m.out_features)


def test_fp6_image_budget_is_honoured_and_a_refusal_is_not_latched(ops):
    """an explicit budget for the kept FP6 images of ALL layers (deploy.fuse(model, fp6_image_budget_bytes=...)); a layer the budget
    refuses runs the transient route and asks again at its next prefill call; release_images() returns its bytes to the account."""
    import flatquant_amd.deploy as deploy
    L4 = deploy.nn.Linear4bit
    gen = torch.Generator().manual_seed(9)
    held0 = L4._fp6_image_bytes_held
    model = torch.nn.Sequential(L4(256, 272), L4(256, 272)).cuda()
    need = int(ops.lib.fq_bf6_blob_bytes(272, 256))
    deploy.fuse(model, fp6_image=True, fp6_image_budget_bytes=held0 + need)       # room for ONE image
    rows = L4.fp6_transient_rows + 3
    p = deploy.PackedQuantizedTensor(torch.from_numpy(rand_packed(gen, rows, 256)[0]).cuda().reshape(1, rows, 128),
                                     (torch.rand(1, 1, rows, generator=gen) * 0.05 + 0.001).half().cuda())
    refs = []
    for lin in model:
        lin.fp6_min_out_features = 0
        lin.weight.copy_(torch.from_numpy(rand_packed(gen, 272, 256)[0]))
        lin.weight_scales.copy_((torch.rand(272, 1, generator=gen) * 0.02 + 0.001))
        refs.append(ops.int4_linear(p.quantized_x.reshape(-1, 128), p.scales_x.reshape(-1), lin.weight, lin.weight_scales.reshape(-1).half(),
                                    None).view(1, rows, 272))
    assert torch.equal(model[0](p), refs[0]) and torch.equal(model[1](p), refs[1])
    assert model[0].image_bytes() == need and model[1].image_bytes() == 0         # the second layer was refused (transient route: same bits)
    assert L4._fp6_image_bytes_held == held0 + need
    model[0].release_images()
    assert L4._fp6_image_bytes_held == held0
    assert torch.equal(model[1](p), refs[1]) and model[1].image_bytes() == need   # asked again, now there is room
    model[1].release_images()
    assert L4._fp6_image_bytes_held == held0


@pytest.mark.parametrize("M", [33, 64, 97, 128])
@pytest.mark.parametrize("Ns,K", [([272], 256), ([4096], 512), ([1040, 528], 384), ([2064, 4096, 16], 256)])
def test_decode_tile_128x128_bit_identical(ops, M, Ns, K):
    """(round 6) M <= 128 on the FP6 path runs 128 x 128 tiles (two waves, two workgroups per CU): the same bits as the int8-path kernel —
    ragged feature counts (a last tile that is mostly padding, a 16-wide problem), one and several problems per launch."""
    gen = torch.Generator().manual_seed(M + sum(Ns) + K)
    probs, refs = [], []
    for N in Ns:
        x, w = torch.from_numpy(rand_packed(gen, M, K)[0]).cuda(), torch.from_numpy(rand_packed(gen, N, K)[0]).cuda()
        sx = (torch.rand(M, generator=gen) * 0.05 + 0.001).half().cuda()
        sw = (torch.rand(N, generator=gen) * 0.02 + 0.0005).half().cuda()
        b = torch.randn(N, generator=gen).half().cuda() if N % 32 == 16 else None
        probs.append((x, sx, w, ops.int4_to_bf6(w, weights=True), sw, b))
        refs.append(ops.int4_linear(x, sx, w, sw, b))
    ys = [ops.int4_linear_fp6(*probs[0])] if len(Ns) == 1 else ops.int4_linear_fp6_multi(probs)
    for y, r in zip(ys, refs):
        assert torch.equal(y.view(torch.int16), r.view(torch.int16))


def test_wide_decode_group_takes_the_fp6_tile_kernel(ops, monkeypatch):
    """linear4bit_multi: a group of >= 16384 output features at 33 .. 128 rows (up + gate at decode batch 64 - 128) runs the FP6 tile kernel
    over the kept images; fewer rows, a narrower group, or a member without a kept image stay on the weight-streaming kernel. Same bits."""
    from flatquant_amd.deploy import PackedQuantizedTensor
    from flatquant_amd.deploy.nn.linear import Linear4bit, linear4bit_multi
    gen = torch.Generator().manual_seed(4)
    K = 256
    mods = []
    for N in (8192, 8192):
        m = Linear4bit(K, N).cuda()
        m.weight.copy_(torch.from_numpy(rand_packed(gen, N, K)[0]))
        m.weight_scales.copy_((torch.rand(N, 1, generator=gen) * 0.02 + 0.0005))
        mods.append(m)
    calls = []
    real_fp6, real_sk = ops.int4_linear_fp6_multi, ops.int4_skinny_linear_multi
    monkeypatch.setattr(ops, "int4_linear_fp6_multi", lambda p: (calls.append("fp6"), real_fp6(p))[1])
    monkeypatch.setattr(ops, "int4_skinny_linear_multi", lambda p: (calls.append("skinny"), real_sk(p))[1])
    for rows, want in ((64, "fp6"), (128, "fp6"), (33, "fp6"), (32, "skinny"), (1, "skinny")):
        xs = [PackedQuantizedTensor(torch.from_numpy(rand_packed(gen, rows, K)[0]).cuda().reshape(1, rows, K // 2),
                                    (torch.rand(1, 1, rows, generator=gen) * 0.05 + 0.001).half().cuda()) for _ in mods]
        calls.clear()
        ys = linear4bit_multi(mods, xs)
        assert calls == [want], (rows, calls)
        for m, x, y in zip(mods, xs, ys):
            ref = ops.int4_linear(x.quantized_x.reshape(rows, -1), x.scales_x.reshape(-1), m.weight, m.weight_scales.reshape(-1).half(), None)
            assert torch.equal(y.reshape(rows, -1).view(torch.int16), ref.view(torch.int16)), rows
    mods[1].fp6_image = False                      # no kept image on one member: the weight-streaming kernel
    mods[1].release_images()
    calls.clear()
    linear4bit_multi(mods, xs[:2] if rows != 1 else xs)
    xs = [PackedQuantizedTensor(torch.from_numpy(rand_packed(gen, 64, K)[0]).cuda().reshape(1, 64, K // 2),
                                (torch.rand(1, 1, 64, generator=gen) * 0.05 + 0.001).half().cuda()) for _ in mods]
    calls.clear()
    linear4bit_multi(mods, xs)
    assert calls == ["skinny"]
