from .kv_cache import (MultiLayerPagedKVCache4Bit, append_kv_f16, append_kv_i4, asym_quantize_and_pack_i4,  # noqa: F401
                       batch_decode_f16, batch_decode_i4, init_kv_f16, init_kv_i4, transform_quantize_kv,
                       unpack_i4_and_asym_dequantize)
