"""flatquant_amd — MI355X (gfx950) implementation of FlatQuant's online-transform + INT4 activation
quantisation hot path, behind the reference's own module surface.

  flatquant_amd.flatquant.*   mirrors flatquant/{flat_utils,quant_utils,trans_utils,flat_linear,
                              function_utils,hadamard_utils}.py   (fake-quant, "path A" API)
  flatquant_amd.deploy.*      mirrors deploy/{__init__,nn,functional}  (real-quant, "path B" API)
  flatquant_amd.ops           tensor-level entry points over the C ABI (include/fqhip.h)
  flatquant_amd.sharding      one-process-per-GPU row sharding + RCCL broadcast of the factor matrices

Importing the package loads flatquant_amd/lib/libfqhip.so and fails loudly if it is missing: there is no
CPU / PyTorch fallback for any op on the path.
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is not built)
from . import ops  # noqa: F401
from . import flatquant  # noqa: F401
from . import deploy  # noqa: F401
from . import sharding  # noqa: F401

__version__ = "0.1.0"
