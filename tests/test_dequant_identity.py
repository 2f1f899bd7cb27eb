"""The identity the Linear4bit epilogue rests on (csrc/fq_gemm_common.hpp, dequant16f): quant.cu:78's int(q / 10.0f) — a float division
truncated toward zero — equals trunc(fl32(q * 0.1f)) AND the C integer division q / 10 for EVERY accumulator value the kernels can
produce (|q| <= 2^24: the fp32 accumulator of the FP6 path holds exact integers up to there, K <= 2^18). Exhaustive, IEEE fp32 (numpy)."""
import numpy as np


def test_trunc_of_the_float_product_is_the_integer_division_for_every_accumulator_value():
    c = np.float32(0.1)
    ten = np.float32(10.0)
    step = 1 << 22
    for lo in range(-(1 << 24), (1 << 24) + 1, step):
        q = np.arange(lo, min(lo + step, (1 << 24) + 1), dtype=np.int64)
        qf = q.astype(np.float32)                                   # exact: |q| <= 2^24
        ref = np.where(q >= 0, q // 10, -((-q) // 10))              # C division, toward zero
        assert np.array_equal(np.trunc(qf / ten).astype(np.int64), ref)     # the reference's form: int(q / 10.0f)
        assert np.array_equal(np.trunc(qf * c).astype(np.int64), ref)       # the kernel's form: v_mul_f32, v_trunc_f32


def test_the_clamp_cannot_bind_up_to_k_10176():
    """|q| <= 64 K (every INT4 product is at most 64 in magnitude): K <= 10176 keeps |q / 10| <= 65126 < 65176, the bound of the
    reference's clamp (quant.cu:78) — the kernels skip it there (dequant16f<false>)."""
    assert 64 * 10176 // 10 == 65126 < 65176
    assert 64 * (10176 + 128) // 10 > 65176          # the next K the FP6 path takes (K % 128 == 0) can reach the bound
