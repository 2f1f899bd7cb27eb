// fq_gemm_common.hpp — pieces shared by the INT4 GEMM kernels (fq_gemm_i4.hip: int8 matrix path, tile and skinny kernels;
// fq_gemm_bf6.hip: FP6 matrix path): the row permutation of the weight operand, the sym_dequant epilogue, the output
// description and the XCD-aware tile order.
#pragma once
#include "fq_common.hpp"

namespace fqgemm {

// A-operand row (0..31) of a 32-row tile -> the n it holds, so that D's lane (h, .) ends with n = 16 h + reg
__device__ __forceinline__ int prow(int c) { return ((c >> 2) & 1) * 16 + (c & 3) + 4 * (c >> 3); }

typedef int i32x4_t __attribute__((ext_vector_type(4)));
// 16 nibbles (8 bytes) -> 16 signed bytes = 16 * q, as {hi(x.x), lo(x.x), hi(x.y), lo(x.y)}: the same element order on
// both operands, which is all the contraction needs (the int8 matrix path: products carry 256)
__device__ __forceinline__ i32x4_t unpack16(uint2 p) {
    i32x4_t r;
#ifdef FQ_GEMM_NOUNPACK  // measurement build (wrong results): what does the unpack cost?
    r[0] = (int)p.x; r[1] = (int)p.y; r[2] = (int)p.x; r[3] = (int)p.y;
    return r;
#endif
    r[0] = (int)(p.x & 0xF0F0F0F0u);
    r[1] = (int)((p.x << 4) & 0xF0F0F0F0u);
    r[2] = (int)(p.y & 0xF0F0F0F0u);
    r[3] = (int)((p.y << 4) & 0xF0F0F0F0u);
    return r;
}

// quant.cu:5-10,66-85: x = s_row * s_col * half(int(q / 10.0f)) * half(10), fp16 products left to right
__device__ __forceinline__ f16 dequant1(int q, f16 srow, f16 scol) {
    int iv = (int)((float)q / 10.0f);  // C truncation toward zero
    iv = max(-65176, min(65176, iv));
    f16 r = srow * scol;
    r = r * (f16)iv;
    return r * (f16)10.0f;
}

// The same for a lane's 16 consecutive features of one token (the tile kernels' epilogue). Round 3, first form (dequant16, 11 VALU
// per element instead of the ~40 of sixteen dequant1 calls — the epilogue had grown to the cost of the whole K loop: 42.7 M VALU
// against 4.2 M MFMA per 16384 x 4096 x 4096 launch, profiles/r03_gemm_bf6_pmc.txt): integer division by 10 (== int(q / 10.0f) for
// |q| < 2^24), the three fp16 products two elements per instruction (v_pk_mul_f16 rounds each half like v_mul_f16), the 16 column
// scales (and biases) as two 16-byte loads. Replaced by the float-pipeline form below.
// From the accumulator AS A FLOAT (the FP6 path's fp32 accumulator holds the exact integer; the int8 path converts once):
// int(q / 10.0f) == trunc(fl(q * 0.1f)) for EVERY |q| <= 2^24 — checked exhaustively on the CPU (tests/test_dequant_identity.py: numpy, IEEE fp32): 0.1f lies
// 1.5e-8 (relative) above 1/10 and the product's rounding adds at most 0.0625, together < 0.1, the distance of q / 10 from the next
// integer below its magnitude — so the division becomes v_mul_f32 + v_trunc_f32, the clamp one v_med3_f32, and the value never
// leaves the float pipeline: 4 - 5 VALU per element instead of 11 (second session of round 3: the epilogue was 5.9 us of every
// 256 x 256 tile, four tiles per CU at 16384 x 4096 x 4096). Bit-identical to dequant1 (tests/test_gpu_gemm_*.py against the integer oracle).
// (the 16 column scales and biases arrive in registers: the FP6-path kernel loads them under its last MFMAs)
// Two elements per instruction where gfx950 has one: v_pk_mul_f32 for the 0.1, v_pk_add_f32 for the + 0.0 (trunc(-0.3) is -0.0 where
// the int of the reference converts to +0.0; -0.0 + 0.0 = +0.0), v_cvt_pk_f16_f32, v_pk_mul_f16 x 3. CLAMP = false when the caller
// knows |q| <= 651760 (K <= 10176: every product is at most 64): the clamp of quant.cu:78 to +-65176 cannot bind.
template <bool CLAMP>
__device__ __forceinline__ void dequant16f(const f32x16& q, f16 srow, f16x8 s0, f16x8 s1, bool has_bias, f16x8 b0, f16x8 b1,
                                           f16x8& o0, f16x8& o1) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f16x2 sr2 = {srow, srow}, ten2 = {(f16)10.0f, (f16)10.0f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {   // element pairs (2j, 2j + 1) of the 16
        const int e = 2 * j;
        f32x2 t = f32x2{q[e], q[e + 1]} * f32x2{0.1f, 0.1f};   // (rounded to fp32: -ffp-contract=off, and nothing fuses into a truncation)
        t = f32x2{__builtin_truncf(t.x), __builtin_truncf(t.y)} + f32x2{0.0f, 0.0f};
        if (CLAMP) {
            t.x = __builtin_amdgcn_fmed3f(t.x, -65176.0f, 65176.0f);
            t.y = __builtin_amdgcn_fmed3f(t.y, -65176.0f, 65176.0f);
        }
        const f16x2 iv = {(f16)t.x, (f16)t.y};
        const f16x2 sc = e < 8 ? f16x2{s0[e], s0[e + 1]} : f16x2{s1[e - 8], s1[e - 7]};
        f16x2 r = sr2 * sc;
        r = r * iv;
        r = r * ten2;
        if (e < 8) {
            o0[e] = r[0];
            o0[e + 1] = r[1];
        } else {
            o1[e - 8] = r[0];
            o1[e - 7] = r[1];
        }
    }
    if (has_bias) {
        o0 = o0 + b0;
        o1 = o1 + b1;
    }
}

// XCD-aware tile order. Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own 4 MB L2. XCD x
// takes a CONTIGUOUS share of the tile sequence, and the sequence walks 8-feature-tile-wide column blocks row by row,
// so the ~32 workgroups resident on an XCD at a time form a 4 x 8 patch of tiles: 12 distinct operand tiles per K
// stage instead of 64, i.e. most of the operand traffic stays in that XCD's L2 instead of crossing the fabric.
__device__ __forceinline__ bool xcd_tile(int bid, int TM, int TN, int& tm, int& tn) {
    const int T = TM * TN, per = (T + 7) >> 3;
    const int xcd = bid & 7, local = bid >> 3;
    const int L = xcd * per + local;
    if (local >= per || L >= T) return false;
    const int blk = L / (8 * TM), rem = L - blk * 8 * TM;
    const int width = TN - blk * 8 < 8 ? TN - blk * 8 : 8;
    tm = rem / width;
    tn = blk * 8 + (rem - tm * width);
    return true;
}

struct GemmOut {
    int32_t* c;          // [M, N] int32, or nullptr
    f16* y;              // [M, N] fp16 (fused dequant), or nullptr
    const f16* srow;     // [M]  activation scales
    const f16* scol;     // [N]  weight scales
    const f16* bias;     // [N] or nullptr
};

}  // namespace fqgemm
