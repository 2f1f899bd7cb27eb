"""GPU parity for the FQ_GROUP128 epilogue of the workgroup-per-token kernel (csrc/fq_kron_generic.hip, round 3):
ActivationQuantizer(groupsize=128) — one scale per 128 consecutive elements of the transformed token
(vllm_custom/model_executor/layers/quantization/utils/fake_quant_utils.py:72-78) — fused into the transform launch for every
output set, fp16 and bf16, plain and grouped (per-expert clip pairs), instead of a transform launch + a row-quantiser launch.

The quantiser stage is checked BIT FOR BIT: the oracle's group quantiser applied to the transform the SAME launch returns;
the fused launch is also compared with the two-launch composition it replaces (identical by construction: both quantise the
transform rounded to the activation dtype). tests/test_gpu_round2.py::test_group128_scales holds the reference-written
fixtures (vLLM class) for 32x64, 64x64, 56x64, 64x112."""
import numpy as np
import pytest
import torch

from oracle import fq_oracle as O
from tests.conftest import same_bits

pytestmark = pytest.mark.gpu
P, F, T, R16, NC0, Q16, SIG16 = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x400
SIG = (0.9820137619972229, 0.7310585786300049)


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def make(M, N, rows, seed, dtype=torch.float16):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, M * N, generator=gen)
    x[:, ::97] *= 20
    L = torch.randn(M, M, generator=gen) / M ** 0.5
    R = torch.randn(N, N, generator=gen) / N ** 0.5
    return x.to(dtype).cuda(), L.to(dtype).cuda(), R.to(dtype).cuda()


@pytest.mark.parametrize("M,N", [(64, 112), (32, 64), (64, 64), (64, 128), (112, 128), (86, 128), (56, 64), (64, 80)])
@pytest.mark.parametrize("rows", [1, 5, 300])
def test_every_output_set_bit_exact_on_own_transform_fp16(ops, M, N, rows):
    if (M * N) % 128:
        pytest.skip("M*N % 128 != 0")
    x, L, R = make(M, N, rows, M + N + rows)
    for sig in (SIG, (1.0, 1.0), (1e-7, 1e-7)):
        o = ops.kron_quant(x, L, R, [sig], T | P | F | R16, groupsize=128)
        y = o.y.cpu().numpy().astype(np.float32)
        ref = O.quant_outputs(y, sig[0], sig[1], groupsize=128)
        assert o.scale[0].shape == (rows, M * N // 128)
        assert np.array_equal(o.q[0].cpu().numpy(), ref["packed"]), (M, N, rows, sig)
        assert np.array_equal(o.scale[0].cpu().numpy(), ref["scale16"]), (M, N, rows, sig)
        assert same_bits(o.fq[0].cpu().numpy(), ref["fq"]), (M, N, rows, sig)
        # the fake-quant-only launch (FlatQuantizedLinear's contract) and the two-launch composition give the same bytes
        f = ops.kron_quant(x, L, R, [sig], F | R16, groupsize=128)
        assert torch.equal(f.fq[0], o.fq[0])
        if (M, N) != (64, 64):   # (at 64 x 64 the plain transform launch is fq_kron64_kernel: another summation order)
            two = ops._quant_groups_of(ops.kron_quant(x, L, R, flags=T).y, [sig], F | P | R16, 128, ops.FusedOutputs())
            assert torch.equal(two.fq[0], o.fq[0]) and torch.equal(two.q[0], o.q[0])


@pytest.mark.parametrize("M,N", [(64, 112), (32, 64), (112, 128)])
@pytest.mark.parametrize("route", ["lac32", "lowp"])
def test_bf16(ops, M, N, route):
    """DeepSeek-V3's own flow is bf16 (main_dpskv3.py:241,395): fp32 quantiser arithmetic with fp32 clip parameters, everything
    in bf16 otherwise (FQ_QUANT_F16 | FQ_SIG_F16 = "in the activation dtype")."""
    x, L, R = make(M, N, 77, 3 * M + N, torch.bfloat16)
    extra = 0 if route == "lac32" else Q16 | SIG16
    o = ops.kron_quant(x, L, R, [SIG], T | F | R16 | extra, groupsize=128)
    yb = o.y.cpu().view(torch.int16).numpy().view(np.uint16)
    y = O.bf16_from_bits(yb).astype(np.float32)
    ref = O.quant_outputs(y, SIG[0], SIG[1], groupsize=128, quant_f16=bool(extra), sig_f16=bool(extra), lowp="bf16")
    fb = o.fq[0].cpu().view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(fb, O.bf16_bits(ref["fq"])), (M, N, route)


def test_grouped_fake_quant_with_group_scales(ops):
    """The routed-expert stage of DeepSeek-V3 (deepseekv3_utils.py:427-452) with a_groupsize = 128: rows sorted by expert,
    per-expert clip pairs, 128-element scales, fake-quant output — one launch; equal to one launch per expert."""
    x, L, R = make(32, 64, 600, 9)
    offs = torch.tensor([0, 0, 7, 7, 300, 599, 600], dtype=torch.int64, device="cuda")
    G = offs.numel() - 1
    gen = torch.Generator().manual_seed(2)
    smax = (0.5 + 0.5 * torch.rand(G, generator=gen)).cuda()
    smin = (0.3 + 0.7 * torch.rand(G, generator=gen)).cuda()
    o = ops.kron_quant_grouped(x, L, R, offs, smax, smin, F | R16, groupsize=128)
    for g in range(G):
        a, b = int(offs[g]), int(offs[g + 1])
        if a == b:
            continue
        one = ops.kron_quant(x[a:b].contiguous(), L, R, [(float(smax[g]), float(smin[g]))], F | R16, groupsize=128)
        assert torch.equal(o.fq[0][a:b], one.fq[0]), g


def test_refused_without_round_y(ops):
    """The epilogue quantises the ROUNDED transform (what the reference's quantiser module is handed); a launch that asks for the
    fp32-accumulator semantics with an output set the wave kernels do not have is refused by the library, and ops composes."""
    from flatquant_amd import _lib
    x, L, R = make(64, 112, 8, 1)
    q = torch.empty(8, 64 * 112 // 2, dtype=torch.uint8, device="cuda")
    s = torch.empty(8 * 56, dtype=torch.float16, device="cuda")
    import ctypes
    P4, F4 = ctypes.c_void_p * 4, ctypes.c_float * 4
    ws = torch.empty(int(_lib.lib.fq_kron_workspace_bytes(64, 112)), dtype=torch.uint8, device="cuda")
    rc = _lib.lib.fq_kron_quant_f16(x.data_ptr(), L.data_ptr(), R.data_ptr(), None, 8, 64, 112, F4(1, 1, 1, 1), F4(1, 1, 1, 1), 1,
                                    P | _lib.FQ_GROUP128, P4(q.data_ptr()), P4(s.data_ptr()), P4(), None, ws.data_ptr(), ws.numel(),
                                    None)
    assert rc == _lib.FQ_EUNSUPPORTED
