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
    raises.  ``reparameterize`` (offline, weights only, fp64 — flat_linear.py:82-97) is kept in torch: it is
    not on the hot path.
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
            self.sigmoid = nn.Sigmoid()
        self._eval_mode = False

    def apply_wclip(self, weight):
        wmin, wmax = weight.min(1, keepdim=True)[0], weight.max(1, keepdim=True)[0]
        wmax = wmax * self.sigmoid(self.clip_factor_w_max)
        wmin = wmin * self.sigmoid(self.clip_factor_w_min)
        return torch.clamp(weight, min=wmin, max=wmax)

    def apply_trans(self, weight, qa_trans):
        if isinstance(qa_trans, list):
            return kronecker_matmul(weight, qa_trans[0].to(weight), qa_trans[1].to(weight))
        return qa_trans(weight, inv_t=True)

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
        weight = self.linear.weight.data
        ori_dtype = weight.dtype
        weight = weight.to(torch.float64)
        if qa_trans is not None:
            weight = self.apply_trans(weight, qa_trans)
        if self.lwc:
            weight = self.apply_wclip(weight)
        if out_trans is not None:
            weight = out_trans(weight.T).T
        if out_trans is not None and self.linear.bias is not None:
            self.linear.bias.data = out_trans(self.linear.bias.data)
        self.linear.weight.data = weight.to(ori_dtype)
        self._eval_mode = True
