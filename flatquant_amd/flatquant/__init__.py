"""Mirror of the reference's ``flatquant`` package for the inference hot path (eval-mode forward only)."""
from . import flat_linear, flat_utils, function_utils, hadamard_utils, quant_utils, trans_utils  # noqa: F401
from .flat_linear import FlatQuantizedLinear  # noqa: F401
from .flat_utils import kronecker_matmul  # noqa: F401
from .function_utils import get_decompose_dim  # noqa: F401
from .quant_utils import ActivationQuantizer  # noqa: F401
from .trans_utils import (InvDecomposeTransMatrix, InvSingleTransMatrix, SVDDecomposeTransMatrix,  # noqa: F401
                          SVDSingleTransMatrix)
