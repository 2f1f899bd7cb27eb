"""Counterpart of flatquant/flat_utils.py for the hot path: ``kronecker_matmul``."""
import torch

from .. import ops
from .._lib import FQ_OUT_TRANSFORM


def kronecker_matmul(x, hadL, hadR):
    """``x.reshape(-1, L, R) @ hadR`` then ``hadL.T @ .`` == ``x_flat @ kron(hadL, hadR)``.

    Reference: flatquant/flat_utils.py:6-17.  fp16 and bf16 CUDA tensors run the fused HIP kernel (intermediate
    rounded to the activation dtype after the right factor, fp32 accumulation, result in the activation dtype — the
    reference's own rounding sequence; bf16 is what its pipeline feeds on Llama-3 / Qwen / DeepSeek,
    model_utils.py:20, main_dpskv3.py:395).  Other dtypes are only used OFF the hot path by the reference (fp64 weight
    re-parameterisation, flat_linear.py:85; fp32 calibration) and are evaluated with torch.matmul.
    """
    if x.dtype in ops.ACT_DTYPES and x.is_cuda:
        hadL = hadL.to(device=x.device, dtype=x.dtype).contiguous()
        hadR = hadR.to(device=x.device, dtype=x.dtype).contiguous()
        return ops.kron_quant(x.contiguous(), hadL, hadR, flags=FQ_OUT_TRANSFORM).y
    if x.dtype in (torch.float32, torch.float64):  # offline use (weights / calibration), not the hot path
        init_shape = x.shape
        y = x.reshape(-1, hadL.shape[0], hadR.shape[0])
        y = torch.matmul(hadL.T.to(y), torch.matmul(y, hadR.to(y)))
        return y.reshape(init_shape)
    raise TypeError(f"kronecker_matmul: activations must be fp16 or bf16 on a ROCm device (got {x.dtype} on {x.device})")
