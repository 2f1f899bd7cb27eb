"""Counterpart of flatquant/quant_utils.py: the per-token activation fake-quantiser."""
import torch

from .. import ops
from .._lib import FQ_ASYM, FQ_OUT_FAKEQUANT, FQ_QUANT_F16, FQ_SIG_F16


def get_qmin_qmax(bits, sym):
    """quant_utils.py:10-16."""
    if sym:
        q_max = torch.tensor(2 ** (bits - 1) - 1)
        q_min = -q_max - 1
    else:
        q_max, q_min = torch.tensor(2 ** bits - 1), 0
    return q_max, q_min


class ActivationQuantizer(torch.nn.Module):
    """Per-token symmetric fake quantisation, one fused HIP pass (row read once, written once).

    Reference: flatquant/quant_utils.py:48-119.  Same constructor, parameters (``clip_factor_a_max/min``,
    shape (1,), init 4.0 when ``lac``) and attributes (``bits sym lac enable q_max q_min groupsize``).
    Arithmetic: with ``lac`` the reference's type promotion evaluates scale, x/scale and scale*q in fp32
    (fp16 / bf16 extrema x fp32 sigmoid); without it everything stays in the activation's dtype — both are reproduced
    (FQ_QUANT_F16), for fp16 and for bf16 activations (model_utils.py:20 torch_dtype='auto').
    4-bit only: symmetric (the activation quantisers) and asymmetric (``sym=False``: the K / V / Q cache quantisers).
    """

    def __init__(self, bits, sym=False, lac=False, groupsize=-1, clip_ratio=None):
        super().__init__()
        ops.invalidate_on_load(self)   # clip factors loaded later: no stale host copies of their sigmoids
        self.bits = bits
        self.q_max, self.q_min = get_qmin_qmax(bits, sym)
        self.sym = sym
        self.groupsize = groupsize
        # groupsize > 0: flatquant/quant_utils.py:59-60 raises; the vLLM copy of the class
        # (vllm_custom/model_executor/layers/quantization/utils/fake_quant_utils.py:72-78) and the DeepSeek flow
        # (--a_groupsize, main_dpskv3.py:512) reshape to (-1, groupsize) first — that behaviour is provided here.
        if self.groupsize not in (-1, 0) and self.groupsize < 8:
            raise NotImplementedError("activation groups of fewer than 8 elements are not supported")
        self.lac = lac
        self._clip_ratio = clip_ratio
        if self.lac:
            init_value = 4.0
            self.sigmoid = torch.nn.Sigmoid()
            self.clip_factor_a_max = torch.nn.Parameter(torch.ones((1,)) * init_value, requires_grad=True)
            self.clip_factor_a_min = torch.nn.Parameter(torch.ones((1,)) * init_value, requires_grad=True)
        self.enable = True

    def _lowp_params(self, x_dtype):
        """lac whose clip parameters have the ACTIVATION's dtype (a module that was .half()'ed / .bfloat16()'ed as a whole,
        or built under torch.set_default_dtype(bfloat16) as main_dpskv3.py:395 does): extremum x sigmoid stays in that
        dtype in the reference, and so do scale, x / scale and scale * q — the all-low-precision route. Any other pairing
        (fp32 parameters next to an fp16 / bf16 activation — the usual case, flat_linear.py:16 under the fp32 default dtype —
        or fp16 parameters next to a bf16 activation) promotes to fp32."""
        return self.lac and self.clip_factor_a_max.dtype == x_dtype

    def _sig(self, x_dtype=torch.float16):
        if self.lac:
            pd = self.clip_factor_a_max.dtype
            if pd in (torch.float16, torch.bfloat16):   # torch.sigmoid of a 16-bit tensor: evaluated in fp32, rounded to it
                return tuple(float(torch.sigmoid(torch.tensor(ops.host_scalar(p.detach()), dtype=pd)))
                             for p in (self.clip_factor_a_max, self.clip_factor_a_min))
            # fp32 sigmoid on the host; the parameters are read back once per version, not once per call
            return ops.sigmoid_pair(self.clip_factor_a_max.detach(), self.clip_factor_a_min.detach())
        if self._clip_ratio is not None:
            return float(self._clip_ratio), float(self._clip_ratio)
        return 1.0, 1.0

    def forward(self, x):
        if self.bits == 16 or (not self.enable):
            return x
        return self.fake_quant(x)

    def fake_quant(self, x):
        # fp16 or bf16 activations (ops.rowquant picks fq_rowquant_f16 / _bf16); FQ_QUANT_F16 / FQ_SIG_F16 mean "in the
        # activation's dtype"
        flags = FQ_OUT_FAKEQUANT | (0 if self.lac else FQ_QUANT_F16)
        if self._lowp_params(x.dtype):
            flags |= FQ_QUANT_F16 | FQ_SIG_F16
        if not self.lac and self._clip_ratio is not None:
            flags |= FQ_SIG_F16     # fp16 extremum x python float: an fp16 product (quant_utils.py:99-100)
        if not self.sym:
            # quant_utils.py:33-46,109-117: the K / V / Q cache quantisers under --k_asym --v_asym (llama_utils.py:124-132)
            flags = (flags & ~FQ_SIG_F16) | FQ_ASYM   # (the asymmetric kernel always rounds the fp16 route's products)
        if self.groupsize > 0 and x.shape[-1] % self.groupsize:
            raise ValueError(f"last dimension {x.shape[-1]} is not a multiple of groupsize {self.groupsize}")
        xin = x.contiguous().reshape(-1, self.groupsize) if self.groupsize > 0 else x.contiguous()
        if self.bits != 4:
            # any other grid of get_qmin_qmax (quant_utils.py:10-16; --a_bits / --q_bits / --k_bits / --v_bits): the plain kernel,
            # same pinned arithmetic with q_max as an argument (fq_fakequant_bits_*; round 4)
            if not 2 <= int(self.bits) <= 8:
                raise NotImplementedError(f"flatquant_amd: ActivationQuantizer(bits={self.bits}): 2..8 and 16 are on the HIP path")
            fl = flags & (FQ_QUANT_F16 | FQ_SIG_F16 | FQ_ASYM)
            return ops.fakequant_bits(xin, self._sig(x.dtype), int(self.bits), fl).reshape(x.shape)
        return ops.rowquant(xin, [self._sig(x.dtype)], flags).fq[0].reshape(x.shape)
