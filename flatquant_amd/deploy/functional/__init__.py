from .online_trans import get_decompose_dim, get_hadK, kronecker_matmul, matmul_hadU_cuda, quant  # noqa: F401
from .quantization import (asym_quant_dequant, pack_i4, sym_quant_dequant, two_compl, unpack_i4)  # noqa: F401
from . import online_trans, quantization  # noqa: F401
