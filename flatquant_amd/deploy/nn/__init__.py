from .linear import Linear4bit, fused_transform_linear, linear4bit_multi  # noqa: F401
from .normalization import RMSNorm  # noqa: F401
from .online_trans import FusedSequential, OnlineTrans, fused_forward  # noqa: F401
from .quantization import Quantizer  # noqa: F401
