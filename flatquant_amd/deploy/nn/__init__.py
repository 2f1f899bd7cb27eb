from .linear import Linear4bit  # noqa: F401
from .normalization import RMSNorm  # noqa: F401
from .online_trans import FusedSequential, OnlineTrans, fused_forward  # noqa: F401
from .quantization import Quantizer  # noqa: F401
