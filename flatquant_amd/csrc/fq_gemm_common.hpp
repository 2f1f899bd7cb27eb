// fq_gemm_common.hpp — pieces shared by the INT4 GEMM kernels (fq_gemm_i4.hip: int8 matrix path, tile and skinny kernels;
// fq_gemm_bf6.hip: FP6 matrix path): the row permutation of the weight operand, the sym_dequant epilogue, the output
// description and the XCD-aware tile order.
#pragma once
#include "fq_common.hpp"

namespace fqgemm {

// A-operand row (0..31) of a 32-row tile -> the n it holds, so that D's lane (h, .) ends with n = 16 h + reg
__device__ __forceinline__ int prow(int c) { return ((c >> 2) & 1) * 16 + (c & 3) + 4 * (c >> 3); }

// quant.cu:5-10,66-85: x = s_row * s_col * half(int(q / 10.0f)) * half(10), fp16 products left to right
__device__ __forceinline__ f16 dequant1(int q, f16 srow, f16 scol) {
    int iv = (int)((float)q / 10.0f);  // C truncation toward zero
    iv = max(-65176, min(65176, iv));
    f16 r = srow * scol;
    r = r * (f16)iv;
    return r * (f16)10.0f;
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
