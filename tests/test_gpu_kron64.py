"""GPU parity for the fused Kronecker transform + INT4 quantisation at d = 4096 (M = N = 64).

Everything goes through the C ABI (flatquant_amd.ops -> libfqhip.so).  Bars:
  * dyadic fixtures: packed bytes, scales, fake-quant and transformed output BIT-EXACT vs BOTH reference paths;
  * quantise/pack stage: BIT-EXACT vs the oracle applied to the kernel's own transformed activation
    (independent of the order in which the matrix cores add partial products);
  * transform stage vs the oracle: fp16 outputs equal except isolated 1-ulp roundings (MFMA vs exact-sum
    accumulation order), <= 1e-3 relative to the token's max — the north-star tolerance.
"""
import numpy as np
import pytest

from conftest import BOUND37, flip_ok
from tests.conftest import same_bits
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu

P, F, T, R16, NC0, Q16 = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def mismatch(a, b):
    return float(np.mean(np.asarray(a).reshape(-1) != np.asarray(b).reshape(-1)))


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def test_exact_fixture_bit_exact_vs_both_reference_paths(ops, golden):
    g = golden("exact_64x64")
    x, L, Rm = dev(g["x"]), dev(g["L"]), dev(g["R"])
    rows = x.shape[0]
    for ci in range(2):
        sig = [(float(g[f"sig{ci}"][0]), float(g[f"sig{ci}"][1]))]
        o = ops.kron_quant(x, L, Rm, sig, P | NC0)                       # deploy / path-B contract
        assert np.array_equal(o.q[0].cpu().numpy(), g[f"b_packed{ci}"])
        assert np.array_equal(o.scale[0].cpu().numpy(), g[f"b_scale{ci}"])
        o = ops.kron_quant(x, L, Rm, sig, F | R16)                       # FlatQuantizedLinear / path-A contract
        assert same_bits(o.fq[0].cpu().numpy(), g[f"a_fq{ci}"].reshape(rows, -1))
        o = ops.kron_quant(x, L, Rm, sig, T)
        assert np.array_equal(o.y.cpu().numpy(), g[f"a_y{ci}"].reshape(rows, -1))


@pytest.mark.parametrize("name", ["kron_A_64x64", "kron_B_64x64", "edge_64x64"])
def test_quant_stage_bit_exact_given_kernel_transform(ops, golden, name):
    g = golden(name)
    Lk, Rk = ("rand_L", "rand_R") if name.startswith("edge") else ("L", "R")
    x, L, Rm = dev(g["x"]), dev(g[Lk]), dev(g[Rk])
    sigs = [(0.9820137619972229, 0.9820137619972229), (0.9, 0.33), (1.0, 1.0)]
    o = ops.kron_quant(x, L, Rm, sigs, T | P | F | R16)
    y16 = o.y.cpu().numpy()
    for ci, (smax, smin) in enumerate(sigs):
        ref = O.quant_outputs(y16.astype(np.float32), smax, smin)
        assert np.array_equal(o.q[ci].cpu().numpy(), ref["packed"])
        assert np.array_equal(o.scale[ci].cpu().numpy(), ref["scale16"])
        assert same_bits(o.fq[ci].cpu().numpy(), ref["fq"])


@pytest.mark.parametrize("sig", [(0.9820137619972229, 0.9820137619972229),   # no clamp needed: the short quantiser
                                 (0.9, 0.33), (0.5, 0.6),                      # clamping variant
                                 (1.0, 1.0), (0.94, 0.93),
                                 (1e-7, 1e-7), (3e-6, 1.0)])                   # quotients beyond the magic-number range
@pytest.mark.parametrize("name", ["kron_A_64x64", "edge_64x64"])
def test_packed_only_kernel_quant_stage_bit_exact(ops, golden, name, sig):
    """The packed-only specialisation (16-wave workgroups) takes three routes through the quantiser (magic-number
    rounding with / without clamp, true division); each must reproduce the oracle on the kernel's own transform."""
    g = golden(name)
    Lk, Rk = ("rand_L", "rand_R") if name.startswith("edge") else ("L", "R")
    x, L, Rm = dev(g["x"]), dev(g[Lk]), dev(g[Rk])
    y16 = ops.kron_quant(x, L, Rm, flags=T).y.cpu().numpy()
    o = ops.kron_quant(x, L, Rm, [sig], P | R16)
    ref = O.quant_outputs(y16.astype(np.float32), sig[0], sig[1])
    assert np.array_equal(o.q[0].cpu().numpy(), ref["packed"])
    assert np.array_equal(o.scale[0].cpu().numpy(), ref["scale16"])


@pytest.mark.parametrize("name", ["kron_A_64x64", "kron_B_64x64", "edge_64x64"])
def test_transform_vs_oracle(ops, golden, name):
    g = golden(name)
    Lk, Rk = ("rand_L", "rand_R") if name.startswith("edge") else ("L", "R")
    x, L, Rm = dev(g["x"]), dev(g[Lk]), dev(g[Rk])
    y = ops.kron_quant(x, L, Rm, flags=T).y.cpu().numpy()
    y32 = O.kron_transform(g["x"], g[Lk], g[Rk]).reshape(y.shape)
    ref = y32.astype(np.float16)
    finite = np.isfinite(ref.astype(np.float32)).all(axis=1)
    assert mismatch(y[finite], ref[finite]) <= 5e-3
    den = np.abs(y32[finite]).max(axis=1, keepdims=True) + 1e-30
    assert np.max(np.abs(y[finite].astype(np.float32) - y32[finite]) / den) <= 1e-3


@pytest.mark.parametrize("name,flags,kw", [
    ("kron_A_64x64", P, dict()),
    ("kron_A_64x64", P | R16, dict(round_y_f16=True)),
    ("kron_B_64x64", P | NC0, dict(clamp0=False)),
])
def test_packed_vs_oracle_random(ops, golden, name, flags, kw):
    g = golden(name)
    x, L, Rm = dev(g["x"]), dev(g["L"]), dev(g["R"])
    for ci in range(2):
        s = (float(g["sig"][ci][0]), float(g["sig"][ci][1]))
        o = ops.kron_quant(x, L, Rm, [s], flags)
        ref = O.kron_quant(g["x"], g["L"], g["R"], s[0], s[1], **kw)
        q = O.unpack_i4(o.q[0].cpu().numpy())
        assert mismatch(q, ref["q"]) <= 1e-3
        assert np.max(np.abs(q - ref["q"].astype(np.int32))) <= 1
        sg = o.scale[0].cpu().numpy().astype(np.float32)
        assert np.max(np.abs(sg - ref["scale"]) / ref["scale"]) <= 1e-3


def test_vs_reference_path_a_and_b_goldens(ops, golden):
    """Directly against the reference's own outputs (not the oracle): INT4 flip rate <= 1e-3, |dq| <= 1."""
    g = golden("kron_A_64x64")
    x, L, Rm = dev(g["x"]), dev(g["L"]), dev(g["R"])
    for ci in range(2):
        s = (float(g["sig"][ci][0]), float(g["sig"][ci][1]))
        o = ops.kron_quant(x, L, Rm, [s], P | F | R16)
        q = O.unpack_i4(o.q[0].cpu().numpy())
        qa = g[f"a16_lac{ci}_q"].reshape(q.shape).astype(np.int32)
        assert mismatch(q, qa) <= 1e-3 and np.max(np.abs(q - qa)) <= 1
        fq, fa = o.fq[0].cpu().numpy().astype(np.float32), g[f"a16_lac{ci}_fq"].reshape(q.shape).astype(np.float32)
        assert np.max(np.abs(fq - fa) / (np.abs(fa).max(axis=1, keepdims=True))) <= 0.15  # one INT4 step at most
        assert flip_ok(fq, fa, "kron64 fake-quant vs path A golden", BOUND37)
    g = golden("kron_B_64x64")
    x, L, Rm = dev(g["x"]), dev(g["L"]), dev(g["R"])
    for ci in range(3):
        s = (float(g["sig"][ci][0]), float(g["sig"][ci][1]))
        o = ops.kron_quant(x, L, Rm, [s], P | NC0)
        q, qb = O.unpack_i4(o.q[0].cpu().numpy()), O.unpack_i4(g[f"b_packed{ci}"])
        assert mismatch(q, qb) <= 1e-3 and np.max(np.abs(q - qb)) <= 1
        sb = g[f"b_scale{ci}"].astype(np.float32)
        assert np.max(np.abs(o.scale[0].cpu().numpy().astype(np.float32) - sb) / sb) <= 1e-3


def test_diag_scale(ops, golden):
    g = golden("kron_A_diag_64x64")
    x, L, Rm, d = dev(g["x"]), dev(g["L"]), dev(g["R"]), dev(g["diag"])
    y = ops.kron_quant(x, L, Rm, flags=T, diag=d).y.cpu().numpy()
    ref = O.kron_transform(g["x"], g["L"], g["R"], diag16=g["diag"]).reshape(y.shape).astype(np.float16)
    assert mismatch(y, ref) <= 5e-3
    assert mismatch(y, g["y"]) <= 1e-2


def test_multi_clip_equals_single_clip_launches(ops, golden):
    g = golden("kron_A_64x64")
    x, L, Rm = dev(g["x"]), dev(g["L"]), dev(g["R"])
    sigs = [(0.98, 0.97), (0.5, 0.9), (1.0, 1.0), (0.75, 0.25)]
    multi = ops.kron_quant(x, L, Rm, sigs, P)
    for i, s in enumerate(sigs):
        one = ops.kron_quant(x, L, Rm, [s], P)
        assert torch.equal(multi.q[i], one.q[0]) and torch.equal(multi.scale[i], one.scale[0])


@pytest.mark.parametrize("rows", [0, 1, 3, 4, 5, 1023, 2049])
def test_ragged_row_counts(ops, rows):
    gen = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 4096, generator=gen).half()
    L = (torch.randn(64, 64, generator=gen) / 8).half()
    Rm = (torch.randn(64, 64, generator=gen) / 8).half()
    o = ops.kron_quant(x.cuda(), L.cuda(), Rm.cuda(), [(1.0, 1.0)], T | P | R16)
    assert o.q[0].shape == (rows, 2048) and o.scale[0].shape == (rows,)
    if rows:
        n = min(rows, 6)
        ref = O.quant_outputs(o.y[-n:].cpu().numpy().astype(np.float32), 1.0, 1.0)
        assert np.array_equal(o.q[0][-n:].cpu().numpy(), ref["packed"])
        y32 = O.kron_transform(x[-n:].numpy(), L.numpy(), Rm.numpy()).reshape(n, -1)
        assert mismatch(o.y[-n:].cpu().numpy(), y32.astype(np.float16)) <= 5e-3


def test_full_size_properties(ops):
    """BASELINE config C2 (8 x 2048 tokens, d = 4096): size-independent properties + sampled oracle check."""
    rows, d = 8 * 2048, 4096
    gen = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(rows, d, generator=gen, device="cuda", dtype=torch.float16)
    x[:, ::97] *= 20
    L = (torch.randn(64, 64, generator=gen, device="cuda") / 8).half()
    Rm = (torch.randn(64, 64, generator=gen, device="cuda") / 8).half()
    sig = [(0.982, 0.982)]
    a = ops.kron_quant(x, L, Rm, sig, T | P | R16)
    b = ops.kron_quant(x, L, Rm, sig, T | P | R16)
    assert torch.equal(a.q[0], b.q[0]) and torch.equal(a.scale[0], b.scale[0])         # deterministic
    perm = torch.randperm(rows, device="cuda")
    c = ops.kron_quant(x[perm].contiguous(), L, Rm, sig, T | P | R16)
    assert torch.equal(c.q[0], a.q[0][perm]) and torch.equal(c.scale[0], a.scale[0][perm])  # token independence
    # dequantised INT4 reproduces the transform to within half a step wherever it is not clipped
    from flatquant_amd.deploy.functional import unpack_i4
    q = unpack_i4(a.q[0]).float()
    s = a.scale[0].float()[:, None]
    y = a.y.float()
    inside = (y.abs() <= 7 * s)
    # the stored scale is the fp16 rounding of the fp32 scale the kernel divided by: |q| * 2^-11 relative slack
    assert bool(((q * s - y).abs()[inside] <= (0.5 + 8 * 2.0 ** -10) * s.expand_as(y)[inside] + 1e-6).all())
    assert int(q.min()) >= -8 and int(q.max()) <= 7
    # linearity of the transform: T(2x) == 2 T(x) exactly in fp16 (power-of-two scaling commutes with rounding)
    y2 = ops.kron_quant((x * 2).contiguous(), L, Rm, flags=T).y
    ok = torch.isfinite(y2).all(dim=1) & torch.isfinite(a.y).all(dim=1) & (a.y.abs().min(dim=1).values > 1e-3)
    assert torch.equal(y2[ok], (a.y * 2)[ok])
    idx = torch.randperm(rows)[:24]
    ref = O.quant_outputs(a.y[idx.cuda()].cpu().numpy().astype(np.float32), *sig[0])
    assert np.array_equal(a.q[0][idx.cuda()].cpu().numpy(), ref["packed"])
    y32 = O.kron_transform(x[idx.cuda()].cpu().numpy(), L.cpu().numpy(), Rm.cpu().numpy()).reshape(24, -1)
    assert mismatch(a.y[idx.cuda()].cpu().numpy(), y32.astype(np.float16)) <= 5e-3


def test_argument_errors(ops):
    x = torch.randn(4, 4096).half()
    L = torch.randn(64, 64).half()
    with pytest.raises(RuntimeError):
        ops.kron_quant(x, L.cuda(), L.cuda())                       # CPU tensor: no CPU path
    with pytest.raises(TypeError):
        ops.kron_quant(x.cuda().float(), L.cuda(), L.cuda())        # wrong dtype
    with pytest.raises(RuntimeError):
        ops.kron_quant(x.cuda().t().contiguous().t(), L.cuda(), L.cuda())   # non-contiguous
    with pytest.raises(ValueError):
        ops.kron_quant(x.cuda(), L.cuda(), L.cuda(), sigs=[])       # no clip set
    from flatquant_amd._lib import FqError
    with pytest.raises(FqError):
        ops.kron_quant(x.cuda(), L.cuda(), L.cuda(), flags=0)       # no output selected


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_prepared_fragment_image_equals_the_in_kernel_gather(dtype):
    """Round 3: the optional 32 KB workspace of the 64 x 64 pair (fq_kron_prepare_f16 + FQ_WS_PREPARED: one coalesced load per
    thread in the prologue instead of eight 2-byte gathers). Every output set, both element types: bit-identical to the
    launch without a workspace; an unprepared workspace is filled by the call itself; ops.kron_quant sees in-place updates."""
    import ctypes
    from flatquant_amd import ops
    from flatquant_amd._lib import FQ_WS_PREPARED, check, lib
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    fn = lib.fq_kron_quant_bf16 if dtype == "bf16" else lib.fq_kron_quant_f16
    gen = torch.Generator().manual_seed(11)
    rows = 777
    x = torch.randn(rows, 4096, generator=gen).to(td).cuda()
    L = (torch.randn(64, 64, generator=gen) / 8).to(td).cuda()
    R = (torch.randn(64, 64, generator=gen) / 8).to(td).cuda()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nbytes = int(lib.fq_kron_workspace_bytes(64, 64))
    assert nbytes == 32768
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    check(lib.fq_kron_prepare_f16(L.data_ptr(), R.data_ptr(), 64, 64, ws.data_ptr(), nbytes, st))
    smax, smin = (ctypes.c_float * 4)(0.93, 0.8), (ctypes.c_float * 4)(0.9, 0.7)

    def run(flags, workspace, wbytes, nclip=1):
        q = [torch.zeros(rows, 2048, dtype=torch.uint8, device="cuda") for _ in range(nclip)]
        s = [torch.zeros(rows, dtype=td, device="cuda") for _ in range(nclip)]
        f = [torch.zeros(rows, 4096, dtype=td, device="cuda") for _ in range(nclip)]
        y = torch.zeros(rows, 4096, dtype=td, device="cuda")
        qa, sa, fa = (ctypes.c_void_p * 4)(), (ctypes.c_void_p * 4)(), (ctypes.c_void_p * 4)()
        for c in range(nclip):
            qa[c], sa[c], fa[c] = q[c].data_ptr(), s[c].data_ptr(), f[c].data_ptr()
        check(fn(x.data_ptr(), L.data_ptr(), R.data_ptr(), None, rows, 64, 64, smax, smin, nclip, flags, qa, sa, fa,
                 y.data_ptr(), workspace, wbytes, st))
        torch.cuda.synchronize()
        return [t.view(torch.int16) if t.dtype == td else t for t in q + s + f + [y]]

    for flags, nclip in ((P, 1), (P, 2), (F | R16, 1), (T, 1), (P | T | F, 2), (F | T | R16 | Q16, 1)):
        a = run(flags, None, 0, nclip)
        b = run(flags | FQ_WS_PREPARED, ws.data_ptr(), nbytes, nclip)
        ws2 = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        c = run(flags, ws2.data_ptr(), nbytes, nclip)               # not prepared: the call writes the image itself
        for ta, tb, tc in zip(a, b, c):
            assert torch.equal(ta, tb) and torch.equal(ta, tc), (dtype, flags)
        assert torch.equal(ws2, ws)
    # the tensor-level entry: workspace cached per (matrices, version)
    o1 = ops.kron_quant(x, L, R, [(0.93, 0.9)], P)
    o2 = ops.kron_quant(x, L, R, [(0.93, 0.9)], P)
    assert torch.equal(o1.q[0], o2.q[0])
    L.mul_(-1.0)
    o3 = ops.kron_quant(x, L, R, [(0.93, 0.9)], P)
    o4 = ops.kron_quant(x, L.clone(), R.clone(), [(0.93, 0.9)], P)
    assert torch.equal(o3.q[0], o4.q[0]) and not torch.equal(o3.q[0], o1.q[0])


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_multi_job_launch_equals_the_launches_it_replaces(dtype):
    """Round 4: fq_kron_quant_multi_{f16,bf16} — several independent jobs (own activations, own factor pair, own outputs, own row
    count: a layer each) in ONE launch — returns, job for job, the bytes of fq_kron_quant (flat_linear.py:75-80 /
    deploy/nn/online_trans.py:61-99 once per layer). Row counts from empty to several tokens per workgroup; more jobs than CUs."""
    from flatquant_amd import ops
    from flatquant_amd._lib import FQ_NO_CLAMP0, FQ_OUT_PACKED
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    gen = torch.Generator().manual_seed(21)
    for rows_list, flags, sig in [([2048, 1, 0, 5, 300, 2048, 77, 1031], FQ_OUT_PACKED | FQ_NO_CLAMP0, (0.9820137619972229, 0.9820137619972229)),
                                  ([17] * 300, FQ_OUT_PACKED, (0.7, 0.55))]:
        xs, ls, rs = [], [], []
        for j, r in enumerate(rows_list):
            x = torch.randn(r, 4096, generator=gen).to(td)
            if r > 3:
                x[3] = 0
                x[:, ::97] *= 15
            xs.append(x.cuda())
            ls.append((torch.randn(64, 64, generator=gen) / 8).to(td).cuda())
            rs.append((torch.randn(64, 64, generator=gen) / 8).to(td).cuda())
        plan = ops.KronMultiPlan(xs, ls, rs, sig, flags)
        for rep in range(2):    # (the plan is reusable: same table, same outputs)
            q, s = plan.run()
            for j, r in enumerate(rows_list):
                one = ops.kron_quant(xs[j], ls[j], rs[j], [sig], flags)
                assert torch.equal(q[j], one.q[0]), (dtype, j, r)
                assert torch.equal(s[j].view(torch.int16), one.scale[0].view(torch.int16)), (dtype, j, r)


def test_multi_job_argument_errors():
    import ctypes
    from flatquant_amd import _lib
    lib = _lib.lib
    t = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.fq_kron_quant_multi_f16(t.data_ptr(), 2, 4, ctypes.c_float(1.0), ctypes.c_float(1.0), _lib.FQ_OUT_FAKEQUANT, st) == _lib.FQ_EUNSUPPORTED
    assert lib.fq_kron_quant_multi_f16(t.data_ptr(), 0, 4, ctypes.c_float(1.0), ctypes.c_float(1.0), _lib.FQ_OUT_PACKED, st) == _lib.FQ_EINVAL
    jobs = (_lib.FqKronJob * 1)(_lib.FqKronJob(0, 0, 0, 0, 8))
    assert lib.fq_kron_multi_prepare(ctypes.cast(jobs, ctypes.c_void_p), 1, t.data_ptr(), 4096, st) == _lib.FQ_EINVAL   # NULL pointers, rows > 0
    assert lib.fq_kron_multi_prepare(ctypes.cast(jobs, ctypes.c_void_p), 1, t.data_ptr(), 8, st) == _lib.FQ_EINVAL      # table too small
    #  the launch refuses a workgroup count other than the one the table was prepared for (fewer would leave every job's
    # tail tokens unwritten, silently)
    x = torch.zeros(8, 4096, dtype=torch.float16, device="cuda")
    ws = torch.zeros(int(lib.fq_kron_workspace_bytes(64, 64)), dtype=torch.uint8, device="cuda")
    q = torch.zeros(8, 2048, dtype=torch.uint8, device="cuda")
    sc = torch.zeros(8, dtype=torch.float16, device="cuda")
    jobs = (_lib.FqKronJob * 2)(_lib.FqKronJob(x.data_ptr(), ws.data_ptr(), q.data_ptr(), sc.data_ptr(), 8),
                                _lib.FqKronJob(x.data_ptr(), ws.data_ptr(), q.data_ptr(), sc.data_ptr(), 8))
    bpj = lib.fq_kron_multi_prepare(ctypes.cast(jobs, ctypes.c_void_p), 2, t.data_ptr(), 4096, st)
    assert bpj > 1
    one, sig = ctypes.c_float(1.0), ctypes.c_float(1.0)
    assert lib.fq_kron_quant_multi_f16(t.data_ptr(), 2, bpj - 1, one, sig, _lib.FQ_OUT_PACKED | _lib.FQ_WS_PREPARED, st) == _lib.FQ_EINVAL
    assert lib.fq_kron_quant_multi_f16(t.data_ptr(), 2, bpj + 1, one, sig, _lib.FQ_OUT_PACKED | _lib.FQ_WS_PREPARED, st) == _lib.FQ_EINVAL
    assert lib.fq_kron_quant_multi_f16(t.data_ptr(), 2, bpj, one, sig, _lib.FQ_OUT_PACKED | _lib.FQ_WS_PREPARED, st) == 0
    torch.cuda.synchronize()
