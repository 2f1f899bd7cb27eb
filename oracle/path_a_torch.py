"""Torch-CPU restatement ("port") of the reference's fake-quant path A, op for op — TEST INFRASTRUCTURE and the
``cpu_baseline`` leg of bench.py only.

  kronecker_matmul      flatquant/flat_utils.py:6-17     (reshape, x @ hadR, hadL.T @ x)
  get_scale_zero        flatquant/quant_utils.py:85-107  (amax/amin, clamp to 0, lac sigmoid, m/q_max, repeat)
  sym_quant_dequant     flatquant/quant_utils.py:3-7,19-30   (x/scale, round_ste, clamp, scale*q)

It keeps the reference's operator sequence (including the full-size ``scale.repeat``) so that its timing is
representative of the reference's PyTorch CPU path; equivalence to the reference is pinned by
tests/test_oracle_golden.py::test_path_a_torch_port_matches_reference_goldens.
"""
import torch


def kronecker_matmul(x, hadL, hadR):
    init_shape = x.shape
    x = x.reshape(-1, hadL.shape[0], hadR.shape[0])
    x = torch.matmul(x, hadR)
    x = torch.matmul(hadL.T, x)
    return x.reshape(init_shape)


def fake_quant(x, sig, lac=True, bits=4):
    q_max = torch.tensor(2 ** (bits - 1) - 1).to(x)
    init_shape = x.shape
    reshaped_x = x.reshape((-1, x.shape[-1]))
    xmax, xmin = reshaped_x.amax(1, keepdim=True), reshaped_x.amin(1, keepdim=True)
    tmp = torch.zeros_like(xmax)
    xmax, xmin = torch.maximum(xmax, tmp), torch.minimum(xmin, tmp)
    if lac:
        xmax = xmax * torch.tensor([sig[0]], dtype=torch.float32)
        xmin = xmin * torch.tensor([sig[1]], dtype=torch.float32)
    xmax = torch.maximum(torch.abs(xmin), xmax)
    tmp = xmax == 0
    scale = xmax / q_max
    scale[tmp] = 1
    scale = scale.repeat(1, reshaped_x.shape[-1]).reshape(init_shape)
    t = x / scale
    q = torch.clamp((t.round() - t) + t, -(q_max + 1), q_max)   # round_ste, quant_utils.py:3-7 (never returns -0.0)
    return (scale * q).to(x.dtype)


def kron_fakequant(x, hadL, hadR, sig, lac=True):
    with torch.no_grad():
        return fake_quant(kronecker_matmul(x, hadL.to(x), hadR.to(x)), sig, lac)
