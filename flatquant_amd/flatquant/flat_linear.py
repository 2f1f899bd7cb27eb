"""Counterpart of flatquant/flat_linear.py: ``FlatQuantizedLinear`` (inference contract)."""
import torch
import torch.nn as nn

from .flat_utils import kronecker_matmul
from .quant_utils import ActivationQuantizer


class FlatQuantizedLinear(nn.Module):
    """Wraps an ``nn.Linear``; eval forward = per-token INT4 fake-quant of the activation (HIP kernel),
    then the wrapped linear.  Reference: flatquant/flat_linear.py:8-97.

    Same constructor (``args`` needs ``w_bits w_asym a_bits a_asym lac a_groupsize lwc``), same attribute /
    parameter names (``linear``, ``act_quantizer.clip_factor_a_{max,min}``, ``clip_factor_w_{max,min}``).
    Calibration (``_train_forward``: weight quantiser in the loop, flat_linear.py:45-67) is out of scope and
    raises.  ``reparameterize`` (offline, weights only, fp64 — flat_linear.py:82-97) folds the transforms and the
    learnable weight clipping into ``linear`` with plain torch ops: it is not on the hot path.
    """

    def __init__(self, args, linear: nn.Linear):
        super().__init__()
        self.args = args
        self.linear = linear
        self.weight_quantizer = None  # offline GPTQ/RTN weight quantiser: out of scope (gptq_utils.py)
        self.act_quantizer = ActivationQuantizer(bits=args.a_bits, sym=not (args.a_asym), lac=args.lac,
                                                 groupsize=args.a_groupsize)
        self.lwc = args.lwc
        if self.lwc:
            lwc_dim = self.linear.weight.shape[0]
            init_value = 4.0
            self.clip_factor_w_max = nn.Parameter(torch.ones((lwc_dim, 1)) * init_value, requires_grad=True)
            self.clip_factor_w_min = nn.Parameter(torch.ones((lwc_dim, 1)) * init_value, requires_grad=True)
        self._eval_mode = False

    def _ori_forward(self, hidden_states):
        return self.linear(hidden_states)

    def _train_forward(self, hidden_states, qa_trans=None, out_trans=None):
        raise NotImplementedError("flatquant_amd implements the inference path only; call reparameterize() first "
                                  "(calibration = flatquant/train_utils.py is out of scope)")

    def forward(self, hidden_states, qa_trans=None, out_trans=None):
        if not self._eval_mode:
            return self._train_forward(hidden_states, qa_trans=qa_trans, out_trans=out_trans)
        return self._eval_forward(hidden_states)

    def _eval_forward(self, hidden_states):
        x_dtype = hidden_states.dtype
        hidden_states = self.act_quantizer(hidden_states).to(x_dtype)
        return self.linear(hidden_states)

    @torch.no_grad()
    def reparameterize(self, qa_trans=None, out_trans=None):
        """Fold the offline pieces into ``linear`` and switch to the inference forward (flat_linear.py:82-97: the
        input-side transform acts on the weight's input axis with the inverse-transpose, learnable weight clipping
        narrows every output row, ``out_trans`` rotates the output axis and the bias). Done once, in float64, off
        the hot path."""
        lin = self.linear
        w = _fold_input_transform(lin.weight.data.double(), qa_trans)
        if self.lwc:
            w = _clip_rows(w, torch.sigmoid(self.clip_factor_w_max), torch.sigmoid(self.clip_factor_w_min))
        if out_trans is not None:
            w = out_trans(w.t()).t()
            if lin.bias is not None:
                lin.bias.data = out_trans(lin.bias.data)
        lin.weight.data = w.to(lin.weight.dtype)
        self._eval_mode = True


def _fold_input_transform(w64: torch.Tensor, qa_trans) -> torch.Tensor:
    """W <- W . T^{-T} along the input axis. A ``[left, right]`` pair is already the inverse-transposed Kronecker
    factors (the q/k/v and up/gate layers share them, llama_utils.py:85-110); a transform module is asked for them."""
    if qa_trans is None:
        return w64
    if isinstance(qa_trans, (list, tuple)):
        left, right = qa_trans
        return kronecker_matmul(w64, left.to(w64), right.to(w64))
    return qa_trans(w64, inv_t=True)


def _clip_rows(w: torch.Tensor, keep_max: torch.Tensor, keep_min: torch.Tensor) -> torch.Tensor:
    """Learnable weight clipping: every output row is clamped to [min * keep_min, max * keep_max] ([out, 1] factors)."""
    hi = w.amax(dim=1, keepdim=True) * keep_max.to(w)
    lo = w.amin(dim=1, keepdim=True) * keep_min.to(w)
    # torch.clamp(w, min=lo, max=hi) = min(max(w, lo), hi): when lo > hi (a single-signed row whose factors cross) the
    # upper bound wins, as in the reference (flat_linear.py:91)
    return torch.clamp(w, min=lo, max=hi)
