"""Counterpart of flatquant/hadamard_utils.py: online Hadamard rotation.

The non-power-of-two factor matrices (orders 12..172, Sloane's library via QuIP#) are DATA shipped in
flatquant_amd/data/hadk.npz as bit-packed sign matrices (extracted by tools/gen_golden.py), not code.
"""
import functools
import math
import os

import numpy as np
import torch

from .. import ops

_HADK_ORDERS = (172, 156, 140, 108, 60, 52, 36, 28, 40, 20, 12)  # probe order of hadamard_utils.py:7-51
_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "hadk.npz")


def is_pow2(n):
    return (n & (n - 1) == 0) and (n > 0)


@functools.lru_cache(maxsize=None)
def _hadk_table(K):
    with np.load(_DATA) as z:
        bits = np.unpackbits(z[f"had{K}"])[: K * K].reshape(K, K)
    return torch.from_numpy(bits.astype(np.float32) * 2 - 1)


def get_hadK(n, transpose=False):
    """(hadK | None, K) with n = K * 2^p.  Reference: hadamard_utils.py:5-55."""
    for K in _HADK_ORDERS:
        if n % K == 0:
            assert is_pow2(n // K)
            h = _hadk_table(K).clone()
            return (h.T.contiguous() if transpose else h), K
    assert is_pow2(n)
    return None, 1


def get_had_pow2(n, norm=True):
    """Sylvester Hadamard matrix, optionally orthonormal.  Reference: hadamard_utils.py:58-65."""
    assert is_pow2(n)
    had = torch.ones(1, 1)
    while had.shape[0] != n:
        had = torch.cat((torch.cat([had, had], 1), torch.cat([had, -had], 1)), 0)
        if norm:
            had = had / math.sqrt(2)
    return had


def get_had(n, decompose=False):
    """hadamard_utils.py:68-86."""
    hads = []
    if is_pow2(n):
        if decompose:
            pow2 = int(math.log2(n))
            l_dim, r_dim = pow2 // 2, pow2 - pow2 // 2
            hads.extend([get_had_pow2(2 ** l_dim), get_had_pow2(2 ** r_dim)])
        else:
            hads.append(get_had_pow2(n))
    else:
        hadR, K = get_hadK(n)
        hadR = hadR / torch.tensor(n).sqrt()
        if (n // K) > 1:
            hads.append(get_had_pow2(n // K, norm=False))
        hads.append(hadR)
    return hads


def matmul_hadU_cuda(X, hadK, K):
    """hadK @ FWHT(X.view(-1, K, n/K)) / sqrt(n) in one HIP kernel.  Reference: hadamard_utils.py:132-141
    (two launches: third-party fast_hadamard_transform + a batched matmul)."""
    hk = None if K == 1 else hadK.to(device=X.device, dtype=torch.float16).contiguous()
    return ops.hadamard(X.contiguous(), K, hk)


def matmul_hadU(X, transpose=False):
    """Normalised Hadamard transform over the last axis.  Reference: hadamard_utils.py:89-110 (python
    butterfly loop + bmm).  fp16 ROCm tensors run the HIP kernel (fp32 butterflies: at least as accurate as
    the reference's fp16 stages); bf16 and fp32 ROCm tensors (round 5; the reference's function takes any float dtype and its own
    callers feed it fp32 / fp64 weights, hadamard_utils.py:121,128) run the fp32-butterfly kernel on exact fp16 pieces of the input
    (ops.hadamard_wide) and return X's dtype. CPU tensors and fp64 raise: flatquant_amd has no CPU path."""
    n = X.shape[-1]
    hadK, K = get_hadK(n, transpose)
    if X.is_cuda and X.dtype in (torch.bfloat16, torch.float32):
        return ops.hadamard_wide(X, K, None if hadK is None else hadK)
    if not (X.is_cuda and X.dtype == torch.float16):
        raise TypeError("matmul_hadU: fp16 / bf16 / fp32 ROCm tensors only (flatquant_amd has no CPU path)")
    return matmul_hadU_cuda(X, hadK, K)


def matmul_hadUt(X):
    return matmul_hadU(X, transpose=True)


def matmul_hadUt_cuda(X, hadK, K):
    """The reference's version passes an unsupported kwarg (hadamard_utils.py:144-145); here it is simply the
    transform with the transposed factor."""
    return matmul_hadU_cuda(X, None if hadK is None else hadK.T.contiguous(), K)
