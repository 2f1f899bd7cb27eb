"""GPU parity for ActivationQuantizer with bits != 4 (fq_fakequant_bits_f16 / _bf16, csrc/fq_quant.hip; round 4): every case the
reference wrote into tests/golden/act_bits.npz (tools/gen_golden.py bits: bits 8 / 6 / 3, symmetric and asymmetric, lac with fp32
parameters, no lac, clip_ratio, a module cast to the activation dtype; fp16 and bf16) BIT FOR BIT, through the C ABI mirror
(ops.fakequant_bits) and through the module mirror (flatquant_amd.flatquant.quant_utils.ActivationQuantizer)."""
import numpy as np
import pytest
import torch

from oracle import fq_oracle as O
from tests.test_oracle_round4 import _bits_cases, bf, bits_case_args

pytestmark = pytest.mark.gpu
ASYM, Q16, S16 = 0x800, 0x20, 0x400


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def _flags():
    from flatquant_amd import _lib
    assert (_lib.FQ_ASYM, _lib.FQ_QUANT_F16, _lib.FQ_SIG_F16) == (ASYM, Q16, S16)


def _dev(g, key, dtag):
    if dtag == "f16":
        return torch.from_numpy(g[key + "_x"]).cuda()
    return torch.from_numpy(g[key + "_x"].view(np.int16)).cuda().view(torch.bfloat16)


def _same(y, g, key, dtag):
    got = y.view(torch.int16).cpu().numpy().view(np.uint16)
    want = g[key + "_y"].view(np.uint16)
    return np.array_equal(got, want)


def test_every_reference_case_bit_exact_through_the_c_abi(ops, golden):
    _flags()
    g = golden("act_bits")
    n = 0
    for key, nb, sym, name, dtag, cols in _bits_cases(g):
        smax, smin, qf16, sf16 = bits_case_args(g, name, dtag)
        flags = (0 if sym else ASYM) | (Q16 if qf16 else 0) | (S16 if (sf16 and sym and qf16) else 0)
        y = ops.fakequant_bits(_dev(g, key, dtag), (smax, smin), nb, flags)
        assert _same(y, g, key, dtag), key
        n += 1
    assert n == 120


@pytest.mark.parametrize("nb", [8, 6, 3])
@pytest.mark.parametrize("sym", [True, False])
def test_module_mirror_bit_exact(golden, nb, sym):
    """ActivationQuantizer(bits=nb) of the mirror package — same constructor, parameters and call as the reference's — on the
    reference's inputs: lac with fp32 parameters, no lac, clip_ratio, the module cast to the activation dtype."""
    from flatquant_amd.flatquant.quant_utils import ActivationQuantizer
    g = golden("act_bits")
    clip = {"lac32": (4.0, 4.0), "lac32b": (1.7, 0.4), "lac16": (2.1, 0.9)}
    for name, kw in (("lac32", dict(lac=True)), ("lac32b", dict(lac=True)), ("plain", dict(lac=False)),
                     ("ratio", dict(lac=False, clip_ratio=0.83)), ("lac16", dict(lac=True))):
        for dtag, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
            for cols in (128, 520):
                key = f"b{nb}_{'sym' if sym else 'asym'}_{name}_{dtag}_{cols}"
                q = ActivationQuantizer(bits=nb, sym=sym, **kw)
                if name in clip:
                    q.clip_factor_a_max.data.fill_(clip[name][0])
                    q.clip_factor_a_min.data.fill_(clip[name][1])
                if name == "lac16":
                    q = q.to(dt)
                q = q.cuda()
                with torch.no_grad():
                    y = q(_dev(g, key, dtag))
                assert y.dtype == dt and _same(y, g, key, dtag), key


def test_four_bits_equals_the_fast_path_and_edge_rows(ops):
    """bits = 4 through this kernel = fq_rowquant_* with FQ_OUT_FAKEQUANT (the 4-bit kernels), every route; empty input; in place."""
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(300, 1024, generator=gen).half()
    x[:, ::53] *= 15
    x[7] = 0
    x = x.cuda()
    for flags, rq in ((0, 0x02), (Q16, 0x02 | Q16), (Q16 | S16, 0x02 | Q16 | S16), (ASYM, 0x02 | ASYM), (ASYM | Q16, 0x02 | ASYM | Q16)):
        a = ops.fakequant_bits(x, (0.9, 0.8), 4, flags)
        b = ops.rowquant(x, [(0.9, 0.8)], rq).fq[0]
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), flags
    assert ops.fakequant_bits(x[:0], (1.0, 1.0), 8).shape == (0, 1024)
    ref = O.rowquant(x.cpu().numpy(), 1.0, 1.0, bits=8)["fq"]
    assert np.array_equal(ops.fakequant_bits(x, (1.0, 1.0), 8).cpu().numpy().view(np.uint16), ref.view(np.uint16))
    with pytest.raises(Exception):
        ops.fakequant_bits(x, (1.0, 1.0), 9)
