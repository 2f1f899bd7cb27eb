"""Counterpart of flatquant/function_utils.py (the pieces the inference path uses)."""
import math

import numpy as np
import torch


def get_decompose_dim(n):
    """Closest factor pair of n: smallest a >= ceil(sqrt(n)) with a^2 - n a perfect square b^2 gives
    (a - b, a + b).  Reference: flatquant/function_utils.py:11-21 (duplicates in
    deploy/nn/online_trans.py:5-15 and deploy/functional/online_trans.py:8-18)."""
    a = math.isqrt(n)
    if a * a < n:
        a += 1
    while True:
        b2 = a * a - n
        b = math.isqrt(b2)
        if b * b == b2:
            return a - b, a + b
        a += 1


def get_init_scale(w_smax, x_smax, alpha=0.5):
    """flatquant/function_utils.py:7-8."""
    return (w_smax.pow(1 - alpha) / x_smax.pow(alpha)).clamp(min=1e-5)


def get_random_orthg(size):
    """Random orthogonal matrix (QR of a Gaussian, sign-fixed), fp64.  Reference: function_utils.py:24-32."""
    q, r = np.linalg.qr(np.random.randn(size, size))
    return torch.from_numpy(q * np.sign(np.diag(r))[None, :])


get_init_weight = get_random_orthg


def get_inverse(matrix):
    """fp64 inverse cast back (function_utils.py:35-37)."""
    return matrix.double().inverse().to(matrix.dtype)
