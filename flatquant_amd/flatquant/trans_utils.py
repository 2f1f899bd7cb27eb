"""Counterpart of flatquant/trans_utils.py — inference (eval-mode) transforms only.

The reference classes are *training* parametrisations (Cayley-orthogonal SVD factors, trans_utils.py:57-84)
that ``to_eval_mode()`` collapses into plain matrices.  Calibration is out of scope,
so these classes are born in eval mode: they own ``matrix_left / matrix_right`` (+ ``*_inv`` = (P^-1)^T) and
``diag_scale`` with the reference's names, so ``flat_matrices.pth`` / state dicts load unchanged, and their
``forward`` runs the HIP kernel.
"""
import torch
import torch.nn as nn

from .. import ops
from .._lib import FQ_OUT_TRANSFORM
from .flat_utils import kronecker_matmul
from .function_utils import get_init_weight, get_inverse


class _Fp16Cache:
    """Device copies, in the activation's dtype (fp16 / bf16), of a module's (usually fp32) matrices, made once per
    (parameter storage, version, device, dtype): the kernels' fragment workspaces are keyed by the address and version of
    the matrices they were packed from, so handing them a fresh .to(fp16) copy on every call would re-pack on every call
    (and pin every copy in that cache). An entry keeps its SOURCE storage alive: a matrix replaced through ``.data =`` /
    ``load_state_dict(assign=True)`` / ``module.to()`` can then not be given the old address (with version 0) and hit the
    stale copy. ``ops.invalidate_caches()`` clears every instance."""

    _instances = None  # weakref.WeakSet of every cache (created lazily)

    def __init__(self):
        import weakref
        if _Fp16Cache._instances is None:
            _Fp16Cache._instances = weakref.WeakSet()
        _Fp16Cache._instances.add(self)
        self._c = {}

    def get(self, p: torch.Tensor, device, dtype=torch.float16) -> torch.Tensor:
        key = (p.data_ptr(), ops.ver(p), str(device), dtype)
        hit = self._c.get(key)
        if hit is not None:
            return hit[0]
        if len(self._c) > 16:
            self._c.clear()
        copy = p.detach().to(device=device, dtype=dtype).contiguous()
        self._c[key] = (copy, p.untyped_storage())   # the storage object pins the address: it cannot be recycled under the key
        return copy

    @classmethod
    def clear_all(cls):
        for c in list(cls._instances or ()):
            c._c.clear()


class _DecomposeTransBase(nn.Module):
    """Shared eval-mode behaviour of {SVD,Inv}DecomposeTransMatrix (trans_utils.py:85-116, :190-213)."""

    def __init__(self, left_size, right_size, add_diag=False, diag_init_para=None, diag_dtype=None):
        super().__init__()
        left = get_init_weight(left_size).to(torch.get_default_dtype())
        right = get_init_weight(right_size).to(torch.get_default_dtype())
        self.matrix_left = nn.Parameter(left, requires_grad=False)
        self.matrix_right = nn.Parameter(right, requires_grad=False)
        self.matrix_left_inv = nn.Parameter(get_inverse(left).T.contiguous(), requires_grad=False)
        self.matrix_right_inv = nn.Parameter(get_inverse(right).T.contiguous(), requires_grad=False)
        self.add_diag = add_diag
        self.use_diag = True
        if self.add_diag:
            if diag_init_para is None:
                diag_init_para = torch.ones(left_size * right_size, dtype=diag_dtype or torch.get_default_dtype())
            self.diag_scale = nn.Parameter(diag_init_para, requires_grad=False)
        self._eval_mode = True
        self._f16 = _Fp16Cache()
        ops.invalidate_on_load(self)

    def to_eval_mode(self):
        self._eval_mode = True

    def forward(self, inp, inv_t=False):
        left, right = (self.matrix_left_inv, self.matrix_right_inv) if inv_t else (self.matrix_left, self.matrix_right)
        use_diag = self.add_diag and self.use_diag
        if inp.dtype in ops.ACT_DTYPES and inp.is_cuda:
            diag = None
            if use_diag:
                d = self.diag_scale.to(inp)
                # x / d is evaluated as x * (1/d) would NOT match the reference's fp16 division; divide first.
                if inv_t:
                    inp = inp / d
                else:
                    diag = d.contiguous()
            l16, r16 = self._f16.get(left, inp.device, inp.dtype), self._f16.get(right, inp.device, inp.dtype)
            return ops.kron_quant(inp.contiguous(), l16, r16, flags=FQ_OUT_TRANSFORM, diag=diag).y
        if use_diag:
            inp = inp / self.diag_scale.to(inp) if inv_t else inp * self.diag_scale.to(inp)
        return kronecker_matmul(inp, left.to(inp), right.to(inp))

    def __repr__(self):
        return (f"{type(self).__name__}(_eval_mode=True, matrix.shape={tuple(self.matrix_left.shape)}, "
                f"matrix_right.shape={tuple(self.matrix_right.shape)})")


class SVDDecomposeTransMatrix(_DecomposeTransBase):
    """trans_utils.py:57-124 (eval mode).  diag_scale is fp32 as in the reference (:78)."""

    def __init__(self, left_size, right_size, add_diag=False, diag_init_para=None):
        super().__init__(left_size, right_size, add_diag, diag_init_para, diag_dtype=torch.float32)


class InvDecomposeTransMatrix(_DecomposeTransBase):
    """trans_utils.py:170-221 (eval mode)."""


class _SingleTransBase(nn.Module):
    """{SVD,Inv}SingleTransMatrix in eval mode (trans_utils.py:21-46, :136-160): y = x.reshape(-1, n) @ matrix."""

    def __init__(self, size):
        super().__init__()
        m = get_init_weight(size).to(torch.get_default_dtype())
        self.matrix = nn.Parameter(m, requires_grad=False)
        self.matrix_inv_t = nn.Parameter(get_inverse(m).T.contiguous(), requires_grad=False)
        self._eval_mode = True
        self._f16 = _Fp16Cache()
        ops.invalidate_on_load(self)

    def to_eval_mode(self):
        self._eval_mode = True

    def get_matrix(self, inv_t=False):
        return self.matrix_inv_t if inv_t else self.matrix

    def forward(self, inp, inv_t=False):
        init_shape = inp.shape
        n = self.matrix.shape[0]
        rows = inp.numel() // n
        if inp.dtype in ops.ACT_DTYPES and inp.is_cuda and n % 2 == 0 and n <= 64:
            # the activation path (llama_utils.py:275-277: attn_output [.., head_dim, num_heads] over the heads axis):
            # the block-transform kernel with the natural (non-transposed) output = inp.reshape(-1, n) @ matrix in the
            # activation's dtype, R rows of n per launch unit (any R the kernel has that divides the row count); n = any even
            # head count up to 64 (28 / 40: Qwen2.5, Llama-2-13B), fp16 or bf16
            for R in (128, 96, 64, 32):
                if rows % R == 0:
                    m16 = self._f16.get(self.get_matrix(inv_t=inv_t), inp.device, inp.dtype)
                    return ops.block_quant(inp.reshape(-1, R, n).contiguous(), m16, flags=FQ_OUT_TRANSFORM,
                                           transpose_out=False).y.reshape(init_shape)
        if inp.dtype in ops.ACT_DTYPES and inp.is_cuda and n in (64, 128) and rows > 0:
            # n = head_dim (round 4): kcache_trans(q, inv_t=True) / kcache_trans(k) / vcache_trans(v) over [.., heads, head_dim]
            # (llama_utils.py:181-199) — rows of n straight through the matrix pipe, fp16 or bf16
            m16 = self._f16.get(self.get_matrix(inv_t=inv_t), inp.device, inp.dtype)
            return ops.single_trans(inp.contiguous(), m16).reshape(init_shape)
        # everything else (the offline fp64 weight-side use in reparameterize, odd sizes): the reference's own op
        matrix = self.get_matrix(inv_t=inv_t).to(inp)
        return inp.reshape(-1, n).matmul(matrix).reshape(init_shape)

    def __repr__(self):
        return f"{type(self).__name__}(eval_mode=True, matrix.shape={tuple(self.matrix.shape)})"


class SVDSingleTransMatrix(_SingleTransBase):
    """trans_utils.py:8-54 (eval mode)."""


class InvSingleTransMatrix(_SingleTransBase):
    """trans_utils.py:128-167 (eval mode)."""
