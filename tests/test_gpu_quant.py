"""GPU parity for the standalone quantisation kernels (bit-exact: byte/integer results of IEEE arithmetic)."""
import numpy as np
import pytest
from tests.conftest import same_bits
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu
P, F, NC0, Q16 = 0x01, 0x02, 0x10, 0x20


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def rand_x(rows, cols, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, cols, generator=g)
    x[:, ::53] *= 15
    return x.half()


@pytest.mark.parametrize("rows,cols", [(1, 8), (7, 64), (33, 4096), (16, 14336), (5, 28672), (9, 11008), (3, 2056)])
@pytest.mark.parametrize("flags,kw", [(P, {}), (P | NC0, dict(clamp0=False)), (P | Q16, dict(quant_f16=True)),
                                      (F, {}), (F | Q16, dict(quant_f16=True)), (P | F, {})])
def test_rowquant_bit_exact(ops, rows, cols, flags, kw):
    x = rand_x(rows, cols, rows * 131 + cols)
    if rows > 2:
        x[1] = 0
        x[2] = x[2].abs()
    sigs = [(0.982, 0.982), (0.6, 0.9)] if not (flags & Q16) else [(1.0, 1.0), (0.982, 0.95)]
    o = ops.rowquant(x.cuda(), sigs, flags)
    for ci, s in enumerate(sigs):
        ref = O.rowquant(x.numpy(), s[0], s[1], **kw)
        if flags & P:
            assert np.array_equal(o.q[ci].cpu().numpy(), ref["packed"])
            assert np.array_equal(o.scale[ci].cpu().numpy(), ref["scale16"])
        if flags & F:
            assert same_bits(o.fq[ci].cpu().numpy(), ref["fq"])


def test_rowquant_matches_reference_path_a_quantizer(ops, golden):
    """ActivationQuantizer goldens (reference path A run on its own transformed activation)."""
    g = golden("kron_A_112x128")
    rows = g["x"].shape[0]
    y = torch.from_numpy(g["a16_lac0_y"].reshape(rows, -1)).cuda()
    for ci in range(2):
        s = (float(g["sig"][ci][0]), float(g["sig"][ci][1]))
        o = ops.rowquant(y, [s], F)
        assert same_bits(o.fq[0].cpu().numpy(), g[f"a16_lac{ci}_fq"].reshape(rows, -1))
    o = ops.rowquant(y, [(1.0, 1.0)], F | Q16)
    assert same_bits(o.fq[0].cpu().numpy(), g["a16_nolac_fq"].reshape(rows, -1))


@pytest.mark.parametrize("rows,cols", [(1, 2), (5, 63), (4, 64), (17, 4096), (3, 4097), (64, 14336)])
def test_sym_quant_bit_exact(ops, rows, cols):
    x = rand_x(rows, cols, 7 * rows + cols)
    s = (x.float().abs().amax(dim=1) / 7).half()
    s[0] = s[0] * 0.5                                    # force clamping on one row
    q = ops.sym_quant(x.cuda(), s.cuda()).cpu().numpy()
    assert np.array_equal(q, O.sym_quant(x.numpy(), s.numpy()))


def test_sym_dequant_bit_exact(ops):
    g = torch.Generator().manual_seed(3)
    q = torch.randint(-200000, 200000, (37, 200), generator=g, dtype=torch.int32)
    q[0, :4] = torch.tensor([0, 9, -9, 700000])
    sr = (torch.rand(37, generator=g) * 0.1).half()
    sc = (torch.rand(200, generator=g) * 0.1).half()
    x = ops.sym_dequant(q.cuda(), sr.cuda(), sc.cuda()).cpu().numpy()
    ref = O.sym_dequant(q.numpy(), sr.numpy(), sc.numpy())
    assert np.array_equal(x.view(np.uint16), ref.view(np.uint16))


def test_deploy_api_round_trip(ops):
    """deploy.nn.Quantizer -> PackedQuantizedTensor -> unpack * scale ~= x (the module-level contract)."""
    from flatquant_amd import deploy
    from flatquant_amd.deploy.functional import pack_i4, unpack_i4
    x = rand_x(64, 4096, 11).cuda()
    qz = deploy.nn.Quantizer(lac=True).cuda()
    p = qz(x)
    assert isinstance(p, deploy.PackedQuantizedTensor) and p.quantized_x.dtype == torch.uint8
    assert qz(p) is p                                    # already packed: pass-through (quantization.py:14,35)
    q = unpack_i4(p.quantized_x)
    assert torch.equal(pack_i4(q.to(torch.int8)), p.quantized_x)
    ref = O.rowquant(x.cpu().numpy(), *ops.sigmoid_pair_f16(4.0, 4.0), quant_f16=True, sig_f16=True)   # (device semantics)
    assert np.array_equal(p.quantized_x.cpu().numpy(), ref["packed"])
    assert np.array_equal(p.scales_x.cpu().numpy().reshape(-1), ref["scale16"])
    again = deploy.sym_quant(x, p.scales_x.reshape(-1))
    assert torch.equal(again, p.quantized_x)             # Quantizer == scales + sym_quant, as in the reference


# ---- asymmetric fake quantisation (ActivationQuantizer(sym=False): the K / V / Q cache quantisers) ----
ASYM = 0x800
ASYM_CASES = [("lac32", False), ("lac32b", False), ("plain", True), ("ratio", True), ("lac16", True)]


def _asym_sig(g, name):
    if f"{name}_sig" in g:
        return float(g[f"{name}_sig"][0]), float(g[f"{name}_sig"][1])
    return (0.83, 0.83) if name == "ratio" else (1.0, 1.0)


@pytest.mark.parametrize("name,f16", ASYM_CASES)
@pytest.mark.parametrize("cols", [128, 64, 1000, 4096, 10240])
def test_rowquant_asym_matches_reference(ops, golden, name, f16, cols):
    """fq_rowquant_f16(FQ_ASYM) against what the reference's ActivationQuantizer(sym=False) returned: every bit."""
    g = golden("act_asym")
    x, y = g[f"{name}_{cols}_x"], g[f"{name}_{cols}_y"]
    o = ops.rowquant(torch.from_numpy(x).cuda(), [_asym_sig(g, name)], F | ASYM | (Q16 if f16 else 0))
    assert np.array_equal(o.fq[0].cpu().numpy().view(np.uint16), y.view(np.uint16))


@pytest.mark.parametrize("rows,cols", [(1, 8), (3, 16), (5, 24), (130, 128), (1027, 128), (9, 136), (7, 520), (4, 2048),
                                       (3, 2056), (2, 4104), (2, 32768), (4096, 64)])
@pytest.mark.parametrize("f16", [False, True])
def test_rowquant_asym_bit_exact_vs_oracle(ops, rows, cols, f16):
    """every lanes-per-row / vectors-per-lane build (one row per 1..64 lanes, 1 / 4 / 8 cached vectors, re-read rows),
    ragged row counts, three clip sets in one launch"""
    x = rand_x(rows, cols, rows * 17 + cols)
    if rows > 2:
        x[1] = 0
        x[2] = x[2].abs()
    sigs = [(0.982, 0.982), (0.55, 0.9), (1.0, 0.31)]
    o = ops.rowquant(x.cuda(), sigs, F | ASYM | (Q16 if f16 else 0))
    for ci, s in enumerate(sigs):
        ref = O.rowquant_asym(x.numpy(), s[0], s[1], quant_f16=f16)
        assert np.array_equal(o.fq[ci].cpu().numpy().view(np.uint16), ref.view(np.uint16)), (rows, cols, f16, ci)


def test_activation_quantizer_module_asym_and_clip_ratio(ops, golden):
    """The module (flatquant_amd.flatquant.quant_utils.ActivationQuantizer) in the reference's five asymmetric
    configurations and the symmetric clip_ratio one, on the reference's inputs, 3-D like the cache quantisers' callers."""
    from flatquant_amd.flatquant.quant_utils import ActivationQuantizer
    g = golden("act_asym")
    for name, kw in (("lac32", dict(lac=True)), ("lac32b", dict(lac=True)), ("plain", {}), ("ratio", dict(clip_ratio=0.83)),
                     ("lac16", dict(lac=True))):
        q = ActivationQuantizer(bits=4, sym=False, **kw)
        if f"{name}_clip" in g:
            q.clip_factor_a_max.data.fill_(float(g[f"{name}_clip"][0]))
            q.clip_factor_a_min.data.fill_(float(g[f"{name}_clip"][1]))
        q = q.cuda()
        if name == "lac16":
            q = q.half()
        for cols in (128, 4096):
            x, y = g[f"{name}_{cols}_x"], g[f"{name}_{cols}_y"]
            out = q(torch.from_numpy(x).cuda().reshape(2, -1, cols))
            assert out.shape == (2, x.shape[0] // 2, cols) and out.dtype == torch.float16
            assert np.array_equal(out.reshape(-1, cols).cpu().numpy().view(np.uint16), y.view(np.uint16)), (name, cols)
    q = ActivationQuantizer(bits=4, sym=True, clip_ratio=0.83).cuda()
    for cols in (128, 4096):
        out = q(torch.from_numpy(g[f"symratio_{cols}_x"]).cuda())
        assert np.array_equal(out.cpu().numpy().view(np.uint16), g[f"symratio_{cols}_y"].view(np.uint16))


def test_rowquant_asym_flag_errors(ops):
    x = rand_x(4, 128, 1).cuda()
    for flags in (P | ASYM, P | F | ASYM, F | ASYM | NC0):
        with pytest.raises(Exception):
            ops.rowquant(x, [(1.0, 1.0)], flags)


def test_functional_quant_lac_and_input_clip_ratio(ops, golden):
    """deploy.functional.online_trans.quant (deploy/functional/online_trans.py:90-110): with clip factors it is the
    arithmetic of deploy.nn.Quantizer(lac=True) — the reference module's own scales and bytes (quantizer_lac.npz);
    with input_clip_ratio the scales are (max|x| / 7).to(fp16) * ratio and the pack is sym_quant."""
    from flatquant_amd.deploy.functional.online_trans import quant
    g = golden("quantizer_lac")
    for ci in range(3):
        x = torch.from_numpy(g[f"x{ci}"]).cuda()
        p = quant(x, float(g[f"clip{ci}"][0]), float(g[f"clip{ci}"][1]))
        assert np.array_equal(p.scales_x.cpu().numpy().reshape(-1), g[f"scales{ci}"])
        assert np.array_equal(p.quantized_x.cpu().numpy(), g[f"packed{ci}"])
    x = rand_x(33, 4096, 5)
    p = quant(x.cuda(), input_clip_ratio=0.9)
    # the scales are torch ops in the reference's order; fp16 x python float is ONE rounding in torch-ROCm's kernel
    # (v_fma_mixlo_f16) and two (fp32, then fp16) on the CPU: equal except on fp32 products that land on an fp16 tie
    scales = ((x.float().abs().amax(dim=-1, keepdim=True) / 7).half().float() * 0.9).half()
    got = p.scales_x.cpu()
    assert got.shape == scales.shape and got.dtype == torch.float16
    ulp = (got.view(torch.int16).int() - scales.view(torch.int16).int()).abs()
    assert int(ulp.max()) <= 1 and int((ulp != 0).sum()) <= 8
    assert np.array_equal(p.quantized_x.cpu().numpy(), O.sym_quant(x.numpy(), got.numpy().reshape(-1)))
    p1 = quant(x.cuda())
    ref = O.rowquant(x.numpy(), quant_f16=True)
    assert np.array_equal(p1.quantized_x.cpu().numpy(), ref["packed"]) and np.array_equal(p1.scales_x.cpu().numpy().reshape(-1), ref["scale16"])


def test_deploy_quantizer_lac_scales_equal_torch_device_ops(ops):
    """deploy/nn/quantization.py:15-28 evaluated by torch ON THE DEVICE, op for op (the reference never runs this module
    anywhere else: its pack kernel is CUDA): the 0-dim fp32 sigmoid is loaded in the result dtype by torch's device kernels,
    so the extremum is multiplied with the fp16-ROUNDED sigmoid (tools/microbench/sig_f16_probe.py; round-2 ADVICE). The HIP
    launch must give the same fp16 scales bit for bit — this is the live form of tests/golden/quantizer_lac.npz."""
    from flatquant_amd import deploy
    for ci, (cmax, cmin) in enumerate([(4.0, 4.0), (2.3, -0.7), (0.31, 1.9), (1.0, 3.0)]):
        x = rand_x(512, 4096, 40 + ci).cuda()
        qz = deploy.nn.Quantizer(lac=True).cuda()
        qz.clip_factor_a_max.fill_(cmax), qz.clip_factor_a_min.fill_(cmin)
        p = qz(x)
        xmax, xmin = x.amax(1, keepdim=True), x.amin(1, keepdim=True)
        tmp = torch.zeros_like(xmax)
        xmax, xmin = torch.maximum(xmax, tmp), torch.minimum(xmin, tmp)
        xmax = xmax * torch.sigmoid(torch.tensor(cmax).to(x.device))
        xmin = xmin * torch.sigmoid(torch.tensor(cmin).to(x.device))
        xmax = torch.maximum(torch.abs(xmin), xmax)
        scales = (xmax / 7).to(torch.float16)
        assert torch.equal(p.scales_x.reshape(-1), scales.reshape(-1)), ci


def test_input_clip_ratio_is_one_launch_and_matches_torch_device_ops(ops):
    """deploy.nn.Quantizer(input_clip_ratio=r) / functional.quant(input_clip_ratio=r) (quantization.py:30, online_trans.py:106):
    scale = (max|x| / 7).to(fp16) * r — FQ_RATIO_POST, one launch: the fp32 product rounded to fp16 (CPU torch; bit for bit
    against that expression evaluated in fp32 here). torch-ROCm's own mul kernel is one fp16 step off on a few per cent of the
    rows (tools/microbench/dbg_ratio.py: no plain rounding reproduces its pattern): bounded, not imitated. Digits = sym_quant with
    OUR scales bit for bit; shapes as the reference's ([rows, 1] for 2-D, [bsz, 1, seq] for 3-D); an all-zero row keeps scale 0."""
    from flatquant_amd import deploy
    from flatquant_amd.deploy.functional.online_trans import quant
    x = rand_x(2 * 300, 4096, 77).cuda()
    x[5] = 0
    for ratio in (0.9, 0.83, 1.0):
        for shaped in (x, x.reshape(2, 300, 4096)):
            qz = deploy.nn.Quantizer(input_clip_ratio=ratio).cuda()
            p = qz(shaped)
            want = (torch.max(torch.abs(shaped), dim=-1)[0].unsqueeze(1) / 7).to(torch.float16) * ratio
            assert p.scales_x.shape == want.shape and p.scales_x.dtype == torch.float16
            exact = ((torch.max(torch.abs(shaped), dim=-1)[0].unsqueeze(1).float() / 7).half().float() * torch.tensor(ratio, dtype=torch.float32, device="cuda")).half()
            assert torch.equal(p.scales_x, exact)
            ulp = (p.scales_x.view(torch.int16).int() - want.view(torch.int16).int()).abs()
            assert int(ulp.max()) <= 1 and int((ulp != 0).sum()) <= 0.08 * ulp.numel()
            assert float(p.scales_x.reshape(-1)[5]) == 0.0
            assert p.quantized_x.shape == shaped.shape[:-1] + (2048,)
            live = torch.ones(600, dtype=torch.bool, device="cuda")
            live[5] = False
            again = deploy.sym_quant(x[live], p.scales_x.reshape(-1)[live].contiguous())   # (OUR scales: see the docstring)
            assert torch.equal(p.quantized_x.reshape(600, -1)[live], again)
            p2 = quant(shaped, input_clip_ratio=ratio)
            assert torch.equal(p2.quantized_x.reshape(600, -1), p.quantized_x.reshape(600, -1)) and p2.scales_x.shape == want.shape
