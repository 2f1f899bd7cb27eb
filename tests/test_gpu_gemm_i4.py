"""GPU parity for the INT4 x INT4 -> INT32 GEMM and the fused Linear4bit (deploy.matmul / deploy.nn.Linear4bit):
integer work, so the bar is bit-exact against the oracle (oracle/fq_oracle.py:int4_matmul, linear4bit)."""
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
                                   (1000, 1024, 1024), (33, 48, 384), (64, 256, 14336),
                                   (7, 40, 96), (65, 24, 160)])   # the last two: general-shape kernel (K % 128, N % 16)
def test_int4_gemm_bit_exact(ops, M, N, K):
    gen = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    xp, xq = rand_packed(gen, M, K)
    wp, wq = rand_packed(gen, N, K)
    c = ops.int4_matmul(torch.from_numpy(xp).cuda(), torch.from_numpy(wp).cuda()).cpu().numpy()
    ref = xq.astype(np.int64) @ wq.astype(np.int64).T
    assert np.array_equal(c, ref.astype(np.int32))
    assert np.array_equal(O.int4_matmul(xp, wp), ref.astype(np.int32))


def test_int4_gemm_extreme_values_no_overflow(ops):
    """All nibbles -8: every product +64, K = 14336 -> 917504 per output; the kernel's 256x-scaled accumulator peaks
    at 2.3e8 < 2^31."""
    M, N, K = 40, 32, 14336
    xp = O.pack_i4(np.full((M, K), -8, dtype=np.int32))
    wp = O.pack_i4(np.full((N, K), -8, dtype=np.int32))
    c = ops.int4_matmul(torch.from_numpy(xp).cuda(), torch.from_numpy(wp).cuda()).cpu().numpy()
    assert np.all(c == 64 * K)
    wp = O.pack_i4(np.full((N, K), 7, dtype=np.int32))
    c = ops.int4_matmul(torch.from_numpy(xp).cuda(), torch.from_numpy(wp).cuda()).cpu().numpy()
    assert np.all(c == -56 * K)


@pytest.mark.parametrize("M,N,K,bias", [(512, 256, 4096, False), (300, 272, 256, True), (129, 1024, 1024, True),
                                        (9, 24, 160, True)])
def test_fused_linear4bit_equals_matmul_then_dequant(ops, M, N, K, bias):
    gen = torch.Generator().manual_seed(M + N + K)
    xp, _ = rand_packed(gen, M, K)
    wp, _ = rand_packed(gen, N, K)
    sx = (torch.rand(M, generator=gen) * 0.05 + 0.001).half()
    sw = (torch.rand(N, generator=gen) * 0.02 + 0.0005).half()
    b = torch.randn(N, generator=gen).half() if bias else None
    y = ops.int4_linear(torch.from_numpy(xp).cuda(), sx.cuda(), torch.from_numpy(wp).cuda(), sw.cuda(),
                        None if b is None else b.cuda()).cpu().numpy()
    ref = O.linear4bit(xp, sx.numpy(), wp, sw.numpy(), None if b is None else b.numpy())
    assert np.array_equal(y.view(np.uint16), ref.view(np.uint16))
    # and against the two-launch route of the reference
    two = ops.sym_dequant(ops.int4_matmul(torch.from_numpy(xp).cuda(), torch.from_numpy(wp).cuda()), sx.cuda(), sw.cuda())
    if b is not None:
        two = two + b.cuda()
    assert torch.equal(two.cpu(), torch.from_numpy(y))


def test_deploy_linear4bit_module_end_to_end(ops):
    """OnlineTrans (Kronecker transform + INT4 quantisation) -> Linear4bit, as deploy/transformers/modeling_llama.py
    chains them; state-dict names as the reference's."""
    import flatquant_amd.deploy as deploy
    gen = torch.Generator().manual_seed(3)
    lin = torch.nn.Linear(4096, 512, bias=True).half()
    ws = (lin.weight.detach().abs().amax(dim=1, keepdim=True) / 7).half()
    q = deploy.nn.Linear4bit.from_float(lin, ws).cuda()
    assert set(q.state_dict()) == {"weight_scales", "weight", "bias"}
    t = deploy.nn.OnlineTrans(4096, trans="matmul", decompose=True, lac=True).cuda()
    t.left_matrix.copy_(torch.eye(64).half())
    t.right_matrix.copy_(torch.eye(64).half())
    x = torch.randn(2, 33, 4096, generator=gen).half().cuda()
    packed = t(x)
    y = q(packed)
    assert y.shape == (2, 33, 512) and y.dtype == torch.float16
    ref = O.linear4bit(packed.quantized_x.reshape(-1, 2048).cpu().numpy(), packed.scales_x.reshape(-1).cpu().numpy(),
                       q.weight.cpu().numpy(), q.weight_scales.reshape(-1).half().cpu().numpy(), q.bias.detach().cpu().numpy())
    assert np.array_equal(y.reshape(-1, 512).cpu().numpy().view(np.uint16), ref.view(np.uint16))
    # identity transform, 4-bit both sides: the result tracks the fp16 linear layer
    full = torch.nn.functional.linear(x.float(), lin.weight.float().cuda(), lin.bias.float().cuda())
    rel = (y.float() - full).norm() / full.norm()
    assert rel < 0.25
    assert torch.equal(deploy.matmul(packed.quantized_x, q.weight).reshape(-1, 512),
                       ops.int4_matmul(packed.quantized_x.reshape(-1, 2048), q.weight))


@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (8, 1024, 4096), (32, 256, 128), (33, 272, 256), (64, 4096, 1024),
                                   (100, 48, 384), (128, 14336, 4096), (5, 4096, 14336), (17, 40, 64)])
def test_skinny_gemm_bit_exact(ops, M, N, K):
    """decode-sized batches: the weight-streaming kernel on the fragment-order weight image == the integer product"""
    gen = torch.Generator().manual_seed(M * 7 + N * 3 + K + 1)
    xp, xq = rand_packed(gen, M, K)
    wp, wq = rand_packed(gen, N, K)
    img = ops.int4_to_frag(torch.from_numpy(wp).cuda())
    c = ops.int4_skinny_matmul(torch.from_numpy(xp).cuda(), img, N).cpu().numpy()
    ref = xq.astype(np.int64) @ wq.astype(np.int64).T
    assert np.array_equal(c, ref.astype(np.int32))


@pytest.mark.parametrize("M,N,K,bias", [(16, 256, 4096, False), (65, 272, 256, True), (128, 1024, 1024, True),
                                        # lone 2048- to 4096-wide projections of <= 32 rows (two weight blobs of a wave in flight)
                                        (32, 4096, 4096, True), (3, 2048, 2048, True), (1, 4096, 14336, False), (7, 2304, 2048, True)])
def test_skinny_linear_equals_tile_kernel(ops, M, N, K, bias):
    gen = torch.Generator().manual_seed(M + N + K + 2)
    xp, _ = rand_packed(gen, M, K)
    wp, _ = rand_packed(gen, N, K)
    sx = (torch.rand(M, generator=gen) * 0.05 + 0.001).half().cuda()
    sw = (torch.rand(N, generator=gen) * 0.02 + 0.0005).half().cuda()
    b = torch.randn(N, generator=gen).half().cuda() if bias else None
    x, w = torch.from_numpy(xp).cuda(), torch.from_numpy(wp).cuda()
    y = ops.int4_skinny_linear(x, sx, ops.int4_to_frag(w), sw, b, N)
    assert torch.equal(y, ops.int4_linear(x, sx, w, sw, b))


@pytest.mark.parametrize("M,K,Ns,bias", [(1, 4096, (4096, 1024, 1024), (False, False, False)), (16, 4096, (14336, 14336), (True, False)),
                                         (33, 256, (272, 48, 40, 512), (True, True, False, True)), (128, 1024, (1024, 96), (False, True)),
                                         (7, 14336, (4096,), (True,))])
def test_skinny_multi_problem_launch_equals_the_single_launches(ops, M, K, Ns, bias):
    """fq_int4_skinny_linear_multi_f16 (round 5): up to four decode-sized problems sharing M and K as ONE launch, each with its own
    activations, weight image, scales, bias and output — bit for bit the separate fq_int4_skinny_linear_f16 launches (and, through
    test_skinny_linear_equals_tile_kernel, the tile kernel and the integer oracle)."""
    gen = torch.Generator().manual_seed(M * 11 + K + sum(Ns))
    problems, want = [], []
    for N, hb in zip(Ns, bias):
        x = torch.from_numpy(rand_packed(gen, M, K)[0]).cuda()
        w = torch.from_numpy(rand_packed(gen, N, K)[0]).cuda()
        sx = (torch.rand(M, generator=gen) * 0.05 + 0.001).half().cuda()
        sw = (torch.rand(N, generator=gen) * 0.02 + 0.0005).half().cuda()
        b = torch.randn(N, generator=gen).half().cuda() if hb else None
        img = ops.int4_to_frag(w)
        problems.append((x, sx, img, sw, b))
        want.append(ops.int4_skinny_linear(x, sx, img, sw, b, N))
    got = ops.int4_skinny_linear_multi(problems)
    assert len(got) == len(Ns)
    for y, ref, N in zip(got, want, Ns):
        assert y.shape == (M, N) and torch.equal(y, ref)


def test_skinny_multi_refuses_what_it_cannot_run(ops):
    gen = torch.Generator().manual_seed(3)
    x = torch.from_numpy(rand_packed(gen, 4, 256)[0]).cuda()
    w = torch.from_numpy(rand_packed(gen, 64, 256)[0]).cuda()
    s4, s64 = torch.ones(4, dtype=torch.float16, device="cuda"), torch.ones(64, dtype=torch.float16, device="cuda")
    one = (x, s4, ops.int4_to_frag(w), s64, None)
    with pytest.raises(ValueError):
        ops.int4_skinny_linear_multi([one] * 5)
    with pytest.raises(RuntimeError):                      # another M
        ops.int4_skinny_linear_multi([one, (x[:2].contiguous(), s4[:2].contiguous(), one[2], s64, None)])
    big = torch.zeros(129, 128, dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception):                         # M > 128: FQ_EUNSUPPORTED
        ops.int4_skinny_linear_multi([(big, torch.ones(129, dtype=torch.float16, device="cuda"), one[2], s64, None)])


def test_linear4bit_multi_takes_one_skinny_launch_for_decode_batches(ops):
    """deploy.nn.linear.linear4bit_multi at decode sizes: q / k / v modules on their own packed inputs == m(x) each."""
    import flatquant_amd.deploy as deploy
    from flatquant_amd.deploy.nn.linear import linear4bit_multi
    gen = torch.Generator().manual_seed(12)
    mods, ins = [], []
    for n_out, hb in ((512, True), (128, False), (128, True)):
        lin = deploy.nn.Linear4bit(256, n_out, bias=hb).cuda()
        lin.weight.copy_(torch.from_numpy(rand_packed(gen, n_out, 256)[0]))
        lin.weight_scales.copy_((torch.rand(n_out, 1, generator=gen) * 0.02 + 0.001))
        if hb:
            lin.bias.copy_(torch.randn(n_out, generator=gen).half())
        mods.append(lin)
        ins.append(deploy.PackedQuantizedTensor(torch.from_numpy(rand_packed(gen, 6, 256)[0]).cuda().reshape(3, 2, 128),
                                                (torch.rand(3, 1, 2, generator=gen) * 0.05 + 0.001).half().cuda()))
    want = [m(x) for m, x in zip(mods, ins)]
    calls = []
    real = ops.int4_skinny_linear_multi
    ops.int4_skinny_linear_multi = lambda pr: calls.append(len(pr)) or real(pr)
    try:
        got = linear4bit_multi(mods, ins)
    finally:
        ops.int4_skinny_linear_multi = real
    assert calls == [3]
    for y, ref in zip(got, want):
        assert y.shape == ref.shape and torch.equal(y, ref)


def test_module_takes_the_skinny_path_for_decode_batches(ops):
    import flatquant_amd.deploy as deploy
    gen = torch.Generator().manual_seed(9)
    lin = deploy.nn.Linear4bit(512, 384, bias=True).cuda()
    lin.weight.copy_(torch.from_numpy(rand_packed(gen, 384, 512)[0]))
    lin.weight_scales.copy_((torch.rand(384, 1, generator=gen) * 0.02 + 0.001))
    lin.bias.copy_(torch.randn(384, generator=gen).half())
    xp, _ = rand_packed(gen, 4, 512)
    p = deploy.PackedQuantizedTensor(torch.from_numpy(xp).cuda().reshape(4, 1, 256),
                                     (torch.rand(4, 1, 1, generator=gen) * 0.05 + 0.001).half().cuda())
    y = lin(p)
    assert getattr(lin, "_dimg", None) is not None
    ref = ops.int4_linear(p.quantized_x.reshape(-1, 256), p.scales_x.reshape(-1), lin.weight,
                          lin.weight_scales.reshape(-1).half(), lin.bias.half()).view(4, 1, 384)
    assert torch.equal(y, ref)


@pytest.mark.parametrize("M,N,K,bias", [(33, 4096, 4096, False), (64, 4096, 14336, True), (100, 4096, 4096, True), (128, 4096, 14336, False),
                                        (64, 1024, 4096, True), (128, 1000, 8192, False), (40, 2048, 2048, False), (64, 4096, 1024, True),
                                        (16, 4096, 14336, True), (32, 4096, 14336, False), (1, 4096, 28672, False), (8, 4096, 14336, True), (9, 2048, 8192, False)])
def test_skinny_linear_with_the_k_range_split_over_workgroups(ops, M, N, K, bias):
    """fq_int4_skinny_linear_split_f16 (round 6): a lone narrow projection of 33 .. 128 rows with every feature tile's K range over 2 - 4
    workgroups (int32 partial tiles in a workspace, the last arriver sums and de-quantises) — bit for bit the unsplit launch and the tile
    kernel; repeated launches (the counters return to zero), geometries that are not split (too few blobs per workgroup) take the plain launch."""
    from flatquant_amd._lib import lib
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randint(0, 256, (M, K // 2), generator=g, device="cuda", dtype=torch.uint8)
    w = torch.randint(0, 256, (N, K // 2), generator=g, device="cuda", dtype=torch.uint8)
    sx = (torch.rand(M, generator=g, device="cuda") * 0.1 + 0.01).half()
    sw = (torch.rand(N, generator=g, device="cuda") * 0.1 + 0.01).half()
    b = torch.randn(N, generator=g, device="cuda").half() if bias else None
    img = ops.int4_to_frag(w)
    want = ops.int4_skinny_linear(x, sx, img, sw, b, N, split=False)
    assert torch.equal(want, ops.int4_linear(x, sx, w, sw, b))
    nbytes = int(lib.fq_int4_skinny_split_workspace_bytes(M, N, K))
    assert (nbytes > 0) == ((N + 31) // 32 <= 128 and K // 64 >= 64 and (M >= 33 or (K >= 8192 and (M >= 9 or K >= 16384))))   # (two rounds of blobs per workgroup; few rows: a long K only)
    for _ in range(3):
        got = ops.int4_skinny_linear(x, sx, img, sw, b, N)
        assert torch.equal(got, want)
    # rows that are not split, and the workspace contract
    assert int(lib.fq_int4_skinny_split_workspace_bytes(8, N, 4096)) == 0 and int(lib.fq_int4_skinny_split_workspace_bytes(M, 8192, K)) == 0
    if nbytes > 0:
        from flatquant_amd import _lib
        y = torch.empty(M, N, dtype=torch.float16, device="cuda")
        small = torch.zeros(256, dtype=torch.uint8, device="cuda")
        rc = lib.fq_int4_skinny_linear_split_f16(x.data_ptr(), sx.data_ptr(), img.data_ptr(), sw.data_ptr(), None, M, N, K, y.data_ptr(),
                                                 small.data_ptr(), 256, None)
        assert rc == _lib.FQ_EINVAL                                             # a workspace that is too small is refused, not overrun
