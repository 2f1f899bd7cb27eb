"""KV-cache quantisation (fq_kv_quant_f16 / fq_kv_dequant_f16) against outputs of the reference's
asym_quantize_and_pack_i4 / unpack_i4_and_asym_dequantize / K transform (tests/golden/kv_quant.npz) and the oracle.
Integer and fp16-arithmetic stages: bit-exact. The K transform (fp32 accumulation in a different order): <= 1 fp16
step on <= 1 % of the values, and the quantiser bit-exact on the kernel's own transform output."""
import numpy as np
import pytest
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ops():
    from flatquant_amd import ops as _ops
    return _ops


def bits(a):
    return np.asarray(a).view(np.uint16)


@pytest.mark.parametrize("lac", [False, True])
def test_values_vs_reference_golden(ops, golden, lac):
    g = golden("kv_quant")
    tag = "lac" if lac else "plain"
    clip = (float(g["clip"][0]), float(g["clip"][1]))
    q, param = ops.kv_quant(dev(g["x"]), None, clip, lac)
    assert np.array_equal(q.cpu().numpy(), g[f"{tag}_q"])
    assert np.array_equal(bits(param[..., 0:1].cpu().numpy()), bits(g[f"{tag}_scale"]))
    assert np.array_equal(bits(param[..., 1:2].cpu().numpy()), bits(g[f"{tag}_zero"]))
    deq = ops.kv_dequant(q, param, lac)
    assert np.array_equal(bits(deq.cpu().numpy()), bits(g[f"{tag}_deq"]))


@pytest.mark.parametrize("lac", [False, True])
def test_key_transform_then_quant(ops, golden, lac):
    g = golden("kv_quant")
    clip = (float(g["clip"][0]), float(g["clip"][1]))
    q, param, y = ops.kv_quant(dev(g["x"]), dev(g["T"]), clip, lac, return_transformed=True)
    y = y.cpu().numpy()
    ref = g["xT"]
    fin = np.isfinite(ref.astype(np.float32)) & np.isfinite(y.astype(np.float32))
    ka, kb = bits(y).astype(np.int32), bits(ref).astype(np.int32)
    ka, kb = np.where(ka & 0x8000, -(ka & 0x7FFF), ka), np.where(kb & 0x8000, -(kb & 0x7FFF), kb)
    st = np.abs(ka - kb)[fin]
    assert st.max() <= 1 and np.mean(st != 0) <= 1e-2
    rows_ok = np.isfinite(y.astype(np.float32)).all(axis=-1)
    p, s, z, _ = O.kv_asym_quant(y, clip[0], clip[1], lac)
    assert np.array_equal(q.cpu().numpy()[rows_ok], p[rows_ok])
    assert np.array_equal(bits(param[..., 0].cpu().numpy())[rows_ok], bits(s[..., 0])[rows_ok])
    assert np.array_equal(bits(param[..., 1].cpu().numpy())[rows_ok], bits(z[..., 0])[rows_ok])


@pytest.mark.parametrize("rows,hd", [(1, 128), (31, 128), (33, 128), (1000, 128), (129, 64), (5, 64)])
@pytest.mark.parametrize("trans", [False, True])
def test_ragged_rows_and_head_dims(ops, rows, hd, trans):
    rng = np.random.default_rng(rows * 7 + hd)
    x = (rng.standard_normal((rows, hd)) * rng.uniform(0.01, 20, (rows, 1))).astype(np.float16)
    T = (rng.standard_normal((hd, hd)) / np.sqrt(hd)).astype(np.float16)
    for lac in (False, True):
        if trans:
            q, param, y = ops.kv_quant(dev(x), dev(T), (0.98, 0.9), lac, return_transformed=True)
            src = y.cpu().numpy()
            assert np.max(np.abs(src.astype(np.float32) - O.kv_transform(x, T).astype(np.float32))) <= 2e-2 * np.abs(src).max()
        else:
            q, param = ops.kv_quant(dev(x), None, (0.98, 0.9), lac)
            src = x
        p, s, z, _ = O.kv_asym_quant(src, np.float16(0.98), np.float16(0.9), lac)
        assert np.array_equal(q.cpu().numpy(), p)
        assert np.array_equal(bits(param[:, 0].cpu().numpy()), bits(s[:, 0]))
        assert np.array_equal(bits(param[:, 1].cpu().numpy()), bits(z[:, 0]))
        assert np.array_equal(bits(ops.kv_dequant(q, param, lac).cpu().numpy()), bits(O.kv_asym_dequant(p, s, z, lac)))


def test_round_trip_error_bound(ops):
    x = (torch.randn(4096, 128, generator=torch.Generator().manual_seed(0)) * 2).half().cuda()
    q, param = ops.kv_quant(x)
    y = ops.kv_dequant(q, param)
    step = param[:, 0:1].float()
    # half a quantisation step plus the fp16 roundings of x + zero, the quotient and q * scale - zero (measured 0.514)
    assert torch.all((y.float() - x.float()).abs() <= 0.55 * step)


def test_module_mirror(ops, golden):
    import flatquant_amd.deploy.transformers as T
    g = golden("kv_quant")
    x = dev(g["x"])
    cm, cn = torch.tensor(float(g["clip"][0])), torch.tensor(float(g["clip"][1]))
    for lac, tag in ((False, "plain"), (True, "lac")):
        q, s, z = T.asym_quantize_and_pack_i4(x, cm, cn, lac=lac)
        assert q.shape == g[f"{tag}_q"].shape and s.shape == g[f"{tag}_scale"].shape
        assert np.array_equal(q.cpu().numpy(), g[f"{tag}_q"])
        assert np.array_equal(bits(s.cpu().numpy()), bits(g[f"{tag}_scale"]))
        assert np.array_equal(bits(z.cpu().numpy()), bits(g[f"{tag}_zero"]))
        assert np.array_equal(bits(T.unpack_i4_and_asym_dequantize(q, s, z, lac=lac).cpu().numpy()), bits(g[f"{tag}_deq"]))
        fq, _, _ = T.asym_quantize_and_pack_i4(x, cm, cn, lac=lac, quantize=False)
        assert np.array_equal(bits(fq.cpu().numpy()), bits(g[f"{tag}_fq"]))
    kq, kp, vq, vp = T.transform_quantize_kv(x, x, dev(g["T"]))
    assert kq.shape == (2, 9, 4, 64) and kp.shape == (18, 4, 2) and vp.shape == (18, 4, 2)
    assert np.array_equal(vq.cpu().numpy(), g["plain_q"])


def test_errors(ops):
    x = torch.zeros(4, 96, dtype=torch.float16, device="cuda")
    with pytest.raises(Exception):
        ops.kv_quant(x)
    with pytest.raises(ValueError):
        ops.kv_quant(torch.zeros(4, 128, dtype=torch.float16, device="cuda"), torch.zeros(64, 64, dtype=torch.float16, device="cuda"))
    assert ops.kv_quant(torch.zeros(0, 128, dtype=torch.float16, device="cuda"))[0].shape == (0, 64)
