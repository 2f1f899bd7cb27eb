from .kv_cache import (asym_quantize_and_pack_i4, transform_quantize_kv,  # noqa: F401
                       unpack_i4_and_asym_dequantize)
