"""Counterpart of deploy/nn/normalization.py."""
import torch

from ... import ops


class RMSNorm(torch.nn.Module):
    """Root-mean-square normalisation without a weight (folded into the next layer at deployment).
    Reference: deploy/nn/normalization.py:4-23 — fp16 is widened to fp32, x * rsqrt(sum(x^2) / mean_dim + eps), back to
    the input dtype. One HIP launch (fq_rmsnorm_f16) for fp16 CUDA tensors whose last axis is ``mean_dim``; pass the
    module to ``OnlineTrans.forward(x, norm=...)`` to fuse it into the transform + quantisation launch instead."""

    def __init__(self, mean_dim: int, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.mean_dim = mean_dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dtype != torch.float16 or x.shape[-1] != self.mean_dim:
            raise RuntimeError("flatquant_amd.deploy.nn.RMSNorm: fp16 input with last axis == mean_dim expected")
        return ops.rmsnorm(x.contiguous(), self.eps)
