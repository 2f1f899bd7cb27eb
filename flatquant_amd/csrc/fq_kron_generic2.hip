// fq_kron_generic2.hip — second translation unit of the workgroup-per-token Kronecker kernel (fq_kron_fast.hpp): the bf16
// instantiations (path-A surface: kronecker_matmul, {Inv,SVD}DecomposeTransMatrix, the fake-quant contract on bf16 activations),
// the launches with one factor pair per group (fq_kron_quant_grouped_mats_*, deepseekv3_utils.py:443-446) and the group-128
// epilogue instantiations (ActivationQuantizer(groupsize=128), fake_quant_utils.py:72-78). Dispatch of everything else:
// fq_kron_generic.hip.
#include "fq_kron_fast.hpp"

// Grouped launch in which every group has its own factor pair (fq_kron_quant_grouped_mats_{f16,bf16}): left [G, M, M],
// right [G, N, N]; the workspace holds G fragment images. The workgroup-per-token kernel (all output sets) for the factor pairs
// it is instantiated for; -1000 otherwise.
template <typename T>
static int launch_kron_grouped_mats_t(int flags, const T* x, const T* left, const T* right, int64_t rows, int M, int N,
                                      FqQuantOut out, int n_groups, void* workspace, int64_t workspace_bytes, int n_cu,
                                      hipStream_t stream) {
    if (M < 1 || N < 2 || (N & 15) || M > 192 || ((M * N / 2) & 15)) return -1000;
    const int64_t img = fq_kron_generic_workspace_bytes(M, N);
    if (!workspace || workspace_bytes < img * n_groups) return -1001;
    if (!(flags & FQ_WS_PREPARED)) {
        const int rc = fq_launch_kron_prepare((const f16*)left, (const f16*)right, M, N, workspace, stream, n_groups);
        if (rc != 0) return rc;
    }
    flags &= ~FQ_WS_PREPARED;
    out.ws_group_stride = img / 16;
    const int MT = tiles32(M), NT = tiles32(N), KS1 = N / 16;
    const uint4* ws = reinterpret_cast<const uint4*>(workspace);
#define FQ_FG(MT_, NT_, KS1_, W_, OCC_)                                                                                  \
    if (MT == MT_ && NT == NT_ && KS1 == KS1_)                                                                           \
        return launch_fast<MT_, NT_, KS1_, W_, OCC_, false, -1, 0, T, true>(flags, x, ws, (const T*)nullptr, rows, M, N, out, n_cu, stream);
    // (the pairs an expert's hidden / model dimension decomposes into: 32x64 = 2048 DeepSeek-V3 moe_inter, 64x112 = 7168,
    //  64x64, 56x64, 64x128, 64x80, 112x128, 86..96x128)
    FQ_FG(1, 2, 4, 4, 4) FQ_FG(2, 2, 4, 4, 2) FQ_FG(2, 4, 7, 4, 2) FQ_FG(2, 4, 8, 4, 2) FQ_FG(2, 3, 5, 4, 2) FQ_FG(4, 4, 8, 4, 2)
    FQ_FG(3, 4, 8, 4, 2)
#undef FQ_FG
    return -1000;
}

int fq_launch_kron_grouped_mats(int flags, const void* x, const void* left, const void* right, int64_t rows, int M, int N,
                                const FqQuantOut& out, int n_groups, void* workspace, int64_t workspace_bytes, int n_cu,
                                hipStream_t stream) {
    if (flags & FQ_DT_BF16)
        return launch_kron_grouped_mats_t<bf16>(flags & ~FQ_DT_BF16, (const bf16*)x, (const bf16*)left, (const bf16*)right, rows, M,
                                                N, out, n_groups, workspace, workspace_bytes, n_cu, stream);
    return launch_kron_grouped_mats_t<f16>(flags, (const f16*)x, (const f16*)left, (const f16*)right, rows, M, N, out, n_groups,
                                           workspace, workspace_bytes, n_cu, stream);
}

// bf16 activations (the path-A surface: kronecker_matmul, {Inv,SVD}DecomposeTransMatrix, the fake-quant contract): the
// all-output-sets instantiation of the workgroup-per-token kernel for the factor pairs of the supported model families,
// fq_kron_general.hip for every other pair; packed-only launches whose token fits a wave take the bf16 instantiations of the
// wave-per-token kernel (fq_kron_wave.hip). The other packed-only families (trio / duo / tall / compile-time output sets) and the
// SiLU.mul / RMSNorm / post-scale forms are the deploy contract, which is fp16-only in the reference (deploy/kernels/*.py assert it).
int fq_launch_kron_wave_bf16(int flags, const void* x, const void* ws, const void* diag, int64_t rows, int M, int N,
                             const FqQuantOut& out, int n_cu, hipStream_t stream);   // fq_kron_wave.hip

int fq_launch_kron_tiles(int flags, const f16* x, const void* ws, const f16* diag, int64_t rows, int M, int N,
                         const FqQuantOut& out, int n_cu, hipStream_t stream);   // fq_kron_tiles.hip
int fq_launch_kron_duo(int flags, const f16* x, const void* ws, const f16* diag, int64_t rows, int M, int N,
                       const FqQuantOut& out, int n_cu, hipStream_t stream);   // fq_kron_duo.hip
int fq_launch_kron_generic_bf16(int flags, const f16* x_, const f16* left_, const f16* right_, const f16* diag_, int64_t rows, int M, int N,
                                const FqQuantOut& out, void* workspace, int64_t workspace_bytes, int n_cu, hipStream_t stream) {
    const bf16 *x = (const bf16*)x_, *left = (const bf16*)left_, *right = (const bf16*)right_, *diag = (const bf16*)diag_;
    if (flags & (FQ_IN_SILU_MUL | FQ_IN_RMSNORM)) return -1000;   // (deploy.nn.RMSNorm and the SiLU.mul fusion are fp16 contracts)
    if (out.post_scale != 0.0f) return -1000;
    if ((out.rt_flags & FQ_GROUP128) && !(flags & FQ_ROUND_Y_F16)) return -1000;  // (the group epilogue quantises the rounded transform)
    const int flags_in = flags;
    flags &= ~FQ_NO_WAVE_KERNEL;
    if (!workspace || workspace_bytes < fq_kron_generic_workspace_bytes(M, N)) return -1001;
    const int MT = tiles32(M), NT = tiles32(N), KS1 = (N + 15) / 16;
    uint4* ws = reinterpret_cast<uint4*>(workspace);
    if (!(flags & FQ_WS_PREPARED)) {  // the fragment image is a re-arrangement of 16-bit words: the same kernel for both types
        const int rc = fq_launch_kron_prepare((const f16*)left, (const f16*)right, M, N, workspace, stream);
        if (rc != 0) return rc;
    }
    flags &= ~FQ_WS_PREPARED;
    const bool spec = !(N & 15) && M <= 192 && !((M * N / 2) & 15);
    const bool no_wave = (flags_in & FQ_NO_WAVE_KERNEL) != 0;
    if (spec && !no_wave && !(out.rt_flags & FQ_GROUP128)) {   // one wave per token where a token fits a wave (packed output only)
        const int rc = fq_launch_kron_wave_bf16(flags, x, ws, diag, rows, M, N, out, n_cu, stream);
        if (rc != -1000) return rc;
    }
    if ((spec || N == 148) && !(out.rt_flags & FQ_GROUP128)) {   // (round 4) token groups of NT waves: 112 x 128, 86 x 128, 80 x 112, 128 x 144, 144 x 192 on bf16
        const int rc = fq_launch_kron_tiles(flags | FQ_DT_BF16, (const f16*)x, ws, (const f16*)diag, rows, M, N, out, n_cu, stream);
        if (rc != -1000) return rc;
    }
    if (spec && !(out.rt_flags & FQ_GROUP128)) {   // (round 4) 128 x 224 on bf16: two token groups per CU (fq_kron_duo.hip)
        const int rc = fq_launch_kron_duo(flags | FQ_DT_BF16, (const f16*)x, ws, (const f16*)diag, rows, M, N, out, n_cu, stream);
        if (rc != -1000) return rc;
    }
    if (out.rt_flags & FQ_GROUP128) {
        if (!spec) return -1000;
#define FQ_FBG(MT_, NT_, KS1_, W_, OCC_) \
    if (MT == MT_ && NT == NT_ && KS1 == KS1_) \
        return launch_fast<MT_, NT_, KS1_, W_, OCC_, false, -1, 0, bf16, false, true>(flags, x, ws, diag, rows, M, N, out, n_cu, stream);
        FQ_FBG(1, 2, 4, 4, 4) FQ_FBG(2, 2, 4, 4, 2) FQ_FBG(2, 4, 7, 4, 2) FQ_FBG(2, 4, 8, 4, 2) FQ_FBG(2, 3, 5, 4, 2) FQ_FBG(3, 4, 8, 4, 2)
        FQ_FBG(4, 4, 8, 4, 2)
#undef FQ_FBG
        return -1000;
    }
    if (spec) {
        int rc;
#define FQ_FB(MT_, NT_, KS1_, W_, OCC_)                                                                                       \
    if (MT == MT_ && NT == NT_ && KS1 == KS1_) {                                                                              \
        rc = launch_fast<MT_, NT_, KS1_, W_, OCC_, false, -1, 0, bf16>(flags, x, ws, diag, rows, M, N, out, n_cu, stream);     \
        if (rc != -1000) return rc;                                                                                           \
    }
        FQ_FB(2, 4, 8, 4, 2) FQ_FB(4, 4, 8, 4, 2) FQ_FB(3, 4, 8, 4, 2) FQ_FB(4, 7, 14, 8, 1) FQ_FB(2, 4, 7, 4, 2)
        FQ_FB(1, 2, 4, 4, 4) FQ_FB(2, 2, 4, 4, 2) FQ_FB(2, 3, 5, 4, 2) FQ_FB(4, 5, 9, 8, 1) FQ_FB(3, 4, 7, 4, 2) FQ_FB(1, 2, 3, 4, 4)
        FQ_FB(5, 6, 12, 8, 1) FQ_FB(6, 6, 11, 8, 1)
#undef FQ_FB
    }
    if (out.rt_flags & FQ_GROUP128) return -1000;   // (the general kernel has no group epilogue)
    return fq_launch_kron_general(flags | FQ_DT_BF16, (const f16*)x, ws, (const f16*)diag, rows, M, N, out, n_cu, stream);
}

// The group-128 epilogue (FQ_GROUP128 with FQ_ROUND_Y_F16) on fp16: the pairs with such an instantiation (DeepSeek-V3: 64 x 112
// hidden, 32 x 64 moe_inter; the N = 64, 80, 128 pairs). -1000 otherwise.
int fq_launch_kron_g128_f16(int flags, const f16* x, const uint4* ws, const f16* diag, int64_t rows, int M, int N, const FqQuantOut& out,
                            int n_cu, hipStream_t stream) {
    const int MT = tiles32(M), NT = tiles32(N), KS1 = (N + 15) / 16;
#define FQ_FG1(MT_, NT_, KS1_, W_, OCC_) \
    if (MT == MT_ && NT == NT_ && KS1 == KS1_) \
        return launch_fast<MT_, NT_, KS1_, W_, OCC_, false, -1, 0, f16, false, true>(flags, x, ws, diag, rows, M, N, out, n_cu, stream);
    FQ_FG1(1, 2, 4, 4, 4) FQ_FG1(2, 2, 4, 4, 2) FQ_FG1(2, 4, 7, 4, 2) FQ_FG1(2, 4, 8, 4, 2)
    FQ_FG1(2, 3, 5, 4, 2) FQ_FG1(3, 4, 8, 4, 2) FQ_FG1(4, 4, 8, 4, 2)
#undef FQ_FG1
    return -1000;
}
