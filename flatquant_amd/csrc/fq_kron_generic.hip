// fq_kron_generic.hip — fused Kronecker transform + per-token INT4 quantisation, ONE WORKGROUP PER TOKEN, for the factor
// pairs deployed models use whose token does not fit a wave or whose output set is not packed-only: d = 14336 (112x128),
// 28672 (128x224), 11008 (86x128), 8192 (64x128), 7168 (64x112), 2048 (32x64), 3584 (56x64), 5120 (64x80), 18432 (128x144),
// 27648 (144x192), 29568 (168x176), 8960 (80x112), and — packed-only, true row length as a template parameter — 18944 (128x148)
// (function_utils.py:11-21 factor pairs); plus the fragment-image builder (fq_kron_prepare_kernel) and the launcher that
// picks a kernel for ANY pair (fq_launch_kron_generic): wave-per-token (fq_kron_wave.hip), three token groups per CU
// (fq_kron_trio.hip), this file's kernel, and fq_kron_general.hip for every other pair (run-time N, N % 16 != 0, M > 192).
//
// Same mathematics and fragment chaining as fq_kron64.hip (U = X.R rounded to fp16, Y^T = U^T.L, the C
// fragment of GEMM 1 is the A fragment of GEMM 2), but a token no longer fits one wave's registers, so:
//   * one 4- or 8-wave workgroup owns one token; the token is staged in LDS once (coalesced 16-byte loads, row
//     pitch padded to an odd number of 16-byte chunks -> conflict-free ds_read_b128 A fragments);
//   * wave w computes the 32-column n'-tiles w, w+WAVES, ...: GEMM 1 for its tile over all rows, fp16
//     conversion in registers, GEMM 2 against all of L; its slice of Y stays in registers;
//   * the B-operand fragments of R and L come from a caller-provided WORKSPACE in fragment order
//     (fq_kron_prepare_kernel, launched by the same C-ABI call; <= 170 KB, L2-resident);
//   * per-token max/min: wave all-reduce + 8 floats of LDS; quantised nibbles / fp16 outputs are staged in
//     LDS and written out with full 16-byte coalesced stores.
// This single pass replaces the reference's split path for M > 64 (kron_matmul.py:213-247), which writes the
// fp16 intermediate to HBM and re-reads it (+4 B/element).
#include "fq_kron_fast.hpp"

namespace {

// Workspace layout: rfrag [NT][KS1][64 lanes] uint4, then lfrag [2*MT][MT][64 lanes] uint4.
// blockIdx.y = group (fq_kron_quant_grouped_mats_*: left [G, M, M], right [G, N, N], one image per group; 0 otherwise).
__global__ void fq_kron_prepare_kernel(const f16* __restrict__ left, const f16* __restrict__ right, int M, int N,
                                       int MT, int NT, int KS1, uint4* __restrict__ ws) {
    const int n_r = NT * KS1 * 64, n_l = 2 * MT * MT * 64;
    left += (size_t)blockIdx.y * M * M;
    right += (size_t)blockIdx.y * N * N;
    ws += (size_t)blockIdx.y * (n_r + n_l);
    for (int item = blockIdx.x * blockDim.x + threadIdx.x; item < n_r + n_l; item += gridDim.x * blockDim.x) {
        f16x8 v;
        if (item < n_r) {
            const int f = item >> 6, ln = item & 63, h = ln >> 5, c = ln & 31;
            const int nt = f / KS1, s = f - nt * KS1;
            const int np = ncol(NT, nt, c);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = s * 16 + h * 8 + j;
                v[j] = (n < N && np < N) ? right[n * N + np] : (f16)0.0f;
            }
        } else {
            const int it = item - n_r;
            const int f = it >> 6, ln = it & 63, h = ln >> 5, c = ln & 31;
            const int ks = f / MT, mo = f - ks * MT;
            const int mp = mo * 32 + c;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = (ks >> 1) * 32 + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
                v[j] = (m < M && mp < M) ? left[m * M + mp] : (f16)0.0f;
            }
        }
        ws[item] = __builtin_bit_cast(uint4, v);
    }
}

}  // namespace

int fq_launch_kron_wave(int flags, const f16* x, const void* ws, const f16* diag, int64_t rows, int M, int N,
                        const FqQuantOut& out, int n_cu, hipStream_t stream);  // fq_kron_wave.hip
int fq_launch_kron_trio(int flags, const f16* x, const void* ws, const f16* diag, int64_t rows, int M, int N,
                        const FqQuantOut& out, int n_cu, hipStream_t stream);
int fq_launch_kron_duo(int flags, const f16* x, const void* ws, const f16* diag, int64_t rows, int M, int N,
                       const FqQuantOut& out, int n_cu, hipStream_t stream);
int fq_launch_kron_tiles(int flags, const f16* x, const void* ws, const f16* diag, int64_t rows, int M, int N,
                         const FqQuantOut& out, int n_cu, hipStream_t stream);   // fq_kron_tiles.hip
int fq_launch_kron_tall(int flags, const f16* x, const void* ws, const f16* diag, int64_t rows, int M, int N,
                        const FqQuantOut& out, int n_cu, hipStream_t stream);  // fq_kron_trio.hip

int64_t fq_kron_generic_workspace_bytes(int M, int N) {
    const int MT = tiles32(M), NT = tiles32(N), KS1 = (N + 15) / 16;
    return ((int64_t)NT * KS1 + 2 * (int64_t)MT * MT) * 1024;
}

int fq_launch_kron_prepare(const f16* left, const f16* right, int M, int N, void* workspace, hipStream_t stream, int groups) {
    const int MT = tiles32(M), NT = tiles32(N), KS1 = (N + 15) / 16;
    const int items = (NT * KS1 + 2 * MT * MT) * 64;
    hipLaunchKernelGGL(fq_kron_prepare_kernel, dim3((items + 255) / 256, groups), dim3(256), 0, stream, left, right, M, N, MT,
                       NT, KS1, reinterpret_cast<uint4*>(workspace));
    return (int)hipGetLastError();
}


// fq_kron_generic2.hip: the bf16 instantiations and the group-128 epilogue instantiations of the same kernel
int fq_launch_kron_generic_bf16(int flags, const f16* x, const f16* left, const f16* right, const f16* diag, int64_t rows, int M, int N,
                                const FqQuantOut& out, void* workspace, int64_t workspace_bytes, int n_cu, hipStream_t stream);
int fq_launch_kron_g128_f16(int flags, const f16* x, const uint4* ws, const f16* diag, int64_t rows, int M, int N, const FqQuantOut& out,
                            int n_cu, hipStream_t stream);

int fq_launch_kron_generic(int flags, const f16* x, const f16* left, const f16* right, const f16* diag,
                           int64_t rows, int M, int N, const FqQuantOut& out, void* workspace,
                           int64_t workspace_bytes, int n_cu, hipStream_t stream) {
    if (M < 1 || N < 2 || (N & 1) || M > 256 || N > 256 || (int64_t)M * N > 32768) return -1000;
    if (flags & FQ_DT_BF16)
        return fq_launch_kron_generic_bf16(flags & ~FQ_DT_BF16, x, left, right, diag, rows, M, N, out, workspace, workspace_bytes, n_cu,
                                           stream);
    // the specialised kernels below: N in whole K-steps, M <= 192, 16-byte packed tokens; every other pair: fq_kron_general.hip
    const bool spec = !(N & 15) && M <= 192 && !((M * N / 2) & 15);
    const bool no_wave = (flags & FQ_NO_WAVE_KERNEL) != 0 || out.post_scale != 0.0f;  // (the wave kernels take no post_scale)
    flags &= ~FQ_NO_WAVE_KERNEL;
    if (flags & FQ_IN_SILU_MUL) {  // fused for the down_proj shapes only; said before any workspace complaint
        if (!spec) return -1000;
        const int mt = tiles32(M), nt = tiles32(N), ks = (N + 15) / 16;
        if (!((mt == 4 && nt == 4 && ks == 8) || (mt == 3 && nt == 4 && ks == 8) || (mt == 4 && nt == 7 && ks == 14) ||
              (mt == 4 && nt == 8 && ks == 16)))
            return -1000;
    }
    if (!workspace || workspace_bytes < fq_kron_generic_workspace_bytes(M, N)) return -1001;
    KronGeom g;
    g.M = M;
    g.N = N;
    g.KS1 = (N + 15) / 16;
    g.pitch = (g.KS1 * 2) | 1;
    const int MT = tiles32(M), NT = tiles32(N);
    uint4* ws = reinterpret_cast<uint4*>(workspace);
    int rc = 0;
    if (!(flags & FQ_WS_PREPARED)) {
        rc = fq_launch_kron_prepare(left, right, M, N, workspace, stream);
        if (rc != 0) return rc;
    }
    flags &= ~FQ_WS_PREPARED;
    if (flags & FQ_IN_SILU_MUL) {  // x_up * silu(x_gate) formed while the token is staged: the down_proj shapes (MT >= 3)
        flags &= ~FQ_IN_SILU_MUL;
        if (diag != nullptr || out.in2 == nullptr) return -1000;
#define FQ_FS(MT_, NT_, KS1_, W_, OCC_)                                                                          \
    if (MT == MT_ && NT == NT_ && g.KS1 == KS1_) {                                                              \
        if ((flags & FQ_CT_MASK) == FQ_OUT_PACKED && !fq_measure_env("FQ_KRON_NO_CTF"))                                 \
            return launch_fast<MT_, NT_, KS1_, W_, OCC_, true, FQ_OUT_PACKED>(flags, x, ws, diag, rows, M, N, out, n_cu, stream); \
        if ((flags & FQ_CT_MASK) == (FQ_OUT_PACKED | FQ_QUANT_F16) && (flags & FQ_ROUND_Y_F16))                         \
            return launch_fast<MT_, NT_, KS1_, W_, OCC_, true, FQ_OUT_PACKED | FQ_QUANT_F16>(flags, x, ws, diag, rows, M, N, out, n_cu, stream); \
        return launch_fast<MT_, NT_, KS1_, W_, OCC_, true>(flags, x, ws, diag, rows, M, N, out, n_cu, stream);  \
    }
#ifndef FQ_FS_DECODE_W
#define FQ_FS_DECODE_W 8   // (measurement knob) waves of the decode-sized 112 x 128 SiLU.mul launch; 4 = the prefill build at every size
#endif
        // decode-sized (a handful of tokens, one workgroup each, the chip idle around them): eight waves per token — the four without an
        // n'-tile halve the rounds of the image copy and of the token's loads (a latency chain, not bandwidth)
        if (FQ_FS_DECODE_W != 4 && rows <= 64 && MT == 4 && NT == 4 && g.KS1 == 8 && (flags & FQ_CT_MASK) == FQ_OUT_PACKED)
            return launch_fast<4, 4, 8, FQ_FS_DECODE_W, 1, true, FQ_OUT_PACKED>(flags, x, ws, diag, rows, M, N, out, n_cu, stream);
        FQ_FS(4, 4, 8, 4, 2) FQ_FS(3, 4, 8, 4, 2) FQ_FS(4, 7, 14, 8, 1) FQ_FS(4, 8, 16, 8, 1)
#undef FQ_FS
        return -1000;
    }
    if (flags & FQ_IN_RMSNORM) {   // fused RMSNorm: the wave-per-token kernel (packed output) is the one that has it
        if (!spec || no_wave || diag != nullptr) return -1000;
        return fq_launch_kron_wave(flags, x, ws, diag, rows, M, N, out, n_cu, stream);
    }
    if (spec && !no_wave && !fq_measure_env("FQ_KRON_NO_WAVE")) {  // one wave per token where a token fits a wave (packed output only)
        rc = fq_launch_kron_wave(flags, x, ws, diag, rows, M, N, out, n_cu, stream);
        if (rc != -1000) return rc;
    }
    // per-128-element scales: the wave kernel above (N = 64, packed, fp32 accumulator) or the group epilogue of the
    // workgroup-per-token kernel's all-output-sets instantiations (the transform rounded to the activation dtype)
    const bool g128 = (out.rt_flags & FQ_GROUP128) != 0;
    if (g128 && (!(flags & FQ_ROUND_Y_F16) || !spec)) return -1000;
    if (!g128 && spec && !fq_measure_env("FQ_KRON_NO_TILES")) { // 80 x 112, 86 x 128, 128 x 144, 144 x 192 (round 4): token groups of NT waves
        rc = fq_launch_kron_tiles(flags, x, ws, diag, rows, M, N, out, n_cu, stream);
        if (rc != -1000) return rc;
    }
    if (!g128 && spec && !fq_measure_env("FQ_KRON_NO_TRIO")) {  // 64 < M <= 128, N = 128, packed output: three token groups one phase apart
        rc = fq_launch_kron_trio(flags, x, ws, diag, rows, M, N, out, n_cu, stream);
        if (rc != -1000) return rc;
    }
    if (!g128 && spec && !fq_measure_env("FQ_KRON_NO_DUO")) {   // 96 < M <= 128, N = 224, packed output: two token groups per CU
        rc = fq_launch_kron_duo(flags, x, ws, diag, rows, M, N, out, n_cu, stream);
        if (rc != -1000) return rc;
    }
    if (!g128 && spec && !fq_measure_env("FQ_KRON_NO_TALL")) {  // 64 < M <= 192, N = 64, packed output: a wave per ROW tile (172 x 64: Hadamard 11008)
        rc = fq_launch_kron_tall(flags, x, ws, diag, rows, M, N, out, n_cu, stream);
        if (rc != -1000) return rc;
    }
#ifdef FQ_MEASURE
    if (const char* dbg = getenv("FQ_KRON_DBG")) flags |= atoi(dbg) & 0x7000;  // measurement: ablation bits of the fast kernel
#endif
    if (!g128 && N == 148 && !fq_measure_env("FQ_KRON_NO_TILES")) {   // (round 4) 128 x 148 on token groups of five waves
        rc = fq_launch_kron_tiles(flags, x, ws, diag, rows, M, N, out, n_cu, stream);
        if (rc != -1000) return rc;
    }
    // N % 16 != 0 in the workgroup-per-token kernel: the packed-only launch of 96 < M <= 128, N = 148 (18944 = 128 x 148,
    // Qwen2.5-7B ffn); its other output sets, diag and every other such pair: fq_kron_general.hip
    if (!g128 && N == 148 && MT == 4 && (flags & FQ_CT_MASK) == FQ_OUT_PACKED && diag == nullptr && !((M * N / 2) & 15) &&
        !fq_measure_env("FQ_KRON_NO_CTF"))
        return launch_fast<4, 5, 10, 8, 1, false, FQ_OUT_PACKED, 148>(flags, x, ws, diag, rows, M, N, out, n_cu, stream);
    if (g128) return fq_launch_kron_g128_f16(flags, x, ws, diag, rows, M, N, out, n_cu, stream);   // (fq_kron_generic2.hip)
    if (spec) {
#define FQ_F(MT_, NT_, KS1_, W_, OCC_)                                                                   \
    if (MT == MT_ && NT == NT_ && g.KS1 == KS1_) {                                                       \
        if (!g128 && MT_ >= 3 && (flags & FQ_CT_MASK) == FQ_OUT_PACKED && !fq_measure_env("FQ_KRON_NO_CTF"))    \
            rc = launch_fast<MT_, NT_, KS1_, W_, OCC_, false, (MT_ >= 3 ? FQ_OUT_PACKED : -1)>(flags, x, ws, diag, rows, M, N, out, n_cu, stream); \
        else if (!g128 && MT_ >= 3 && (flags & FQ_CT_MASK) == (FQ_OUT_PACKED | FQ_QUANT_F16) && (flags & FQ_ROUND_Y_F16)) \
            rc = launch_fast<MT_, NT_, KS1_, W_, OCC_, false, (MT_ >= 3 ? (FQ_OUT_PACKED | FQ_QUANT_F16) : -1)>(flags, x, ws, diag, rows, M, N, out, n_cu, stream); \
        else                                                                                             \
            rc = launch_fast<MT_, NT_, KS1_, W_, OCC_>(flags, x, ws, diag, rows, M, N, out, n_cu, stream); \
        if (rc != -1000) return rc;                                                                      \
    }
#ifndef FQ_GEN_OCC2
#define FQ_GEN_OCC2 2   // MT = 2 shapes (measurement knobs): workgroups per CU, waves per workgroup
#endif
#ifndef FQ_GEN_W2
#define FQ_GEN_W2 4
#endif
        FQ_F(2, 4, 8, FQ_GEN_W2, FQ_GEN_OCC2) FQ_F(4, 4, 8, 4, 2) FQ_F(3, 4, 8, 4, 2) FQ_F(4, 7, 14, 8, 1) FQ_F(2, 4, 7, FQ_GEN_W2, FQ_GEN_OCC2)
        FQ_F(4, 8, 16, 8, 1)  // 112 x 256: the Hadamard rotation of 28672 = (28 x 4) x 256 as a Kronecker product
        FQ_F(1, 2, 4, 4, 4) FQ_F(2, 2, 4, 4, FQ_GEN_OCC2) FQ_F(2, 3, 5, 4, FQ_GEN_OCC2)
        FQ_F(4, 5, 9, 8, 1)   // 128 x 144 = 18432 (DeepSeek-V3 dense ffn), and every 96 < M <= 128 with N = 144
        FQ_F(3, 4, 7, 4, 2)   // 80 x 112 = 8960 (Qwen2.5-1.5B ffn)
        FQ_F(1, 2, 3, 4, 4)   // 32 x 48 = 1536 (Qwen2.5-1.5B hidden)
        FQ_F(5, 6, 12, 8, 1)  // 144 x 192 = 27648 (Qwen2.5-32B ffn)
        FQ_F(6, 6, 11, 8, 1)  // 168 x 176 = 29568 (Qwen2.5-72B ffn)
#undef FQ_F
    }
    if (g128) return -1000;   // (the general kernel has no group epilogue)
    return fq_launch_kron_general(flags, x, ws, diag, rows, M, N, out, n_cu, stream);
}
