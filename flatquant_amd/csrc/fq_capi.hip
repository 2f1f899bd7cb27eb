// fq_capi.hip — the C ABI of libfqhip.so (include/fqhip.h). Argument validation + dispatch only.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fq_common.hpp"

// launchers defined in the kernel translation units
int fq_launch_kron64(int flags, const f16* x, const f16* left, const f16* right, const f16* diag,
                     int64_t rows, const FqQuantOut& out, int n_cu, hipStream_t stream, const void* prep = nullptr);
int fq_launch_kron64_prepare(const void* left, const void* right, void* image, hipStream_t stream);
int fq_launch_kron64_linear(const f16* x, const void* prep, int64_t M, float rms_eps, bool rms, int rt_flags, int n, const void* const* wimg,
                            const f16* const* scol, const f16* const* bias, const float* sig_max, const float* sig_min, const int* N,
                            f16* const* y, int n_cu, hipStream_t stream);
// The optional workspace of the 64 x 64 pair: the 16 KB fragment image fq_kron64_kernel reads, followed by the 16 KB image of the
// workgroup-per-token kernel (the few output sets fq_kron64 has no instantiation for fall through to it).
constexpr int64_t FQ_K64_IMAGE_BYTES = 16384, FQ_K64_WS_BYTES = 32768;
int fq_launch_kron_generic(int flags, const f16* x, const f16* left, const f16* right, const f16* diag,
                           int64_t rows, int M, int N, const FqQuantOut& out, void* workspace,
                           int64_t workspace_bytes, int n_cu, hipStream_t stream);
int fq_launch_gemm_bf6_multi(int n, const uint8_t* const* xblob, const uint8_t* const* wblob, int64_t M, const int* Ns, int K, f16* const* y,
                             const f16* const* srow, const f16* const* scol, const f16* const* bias, hipStream_t stream);   // fq_gemm_bf6.hip
int fq_launch_fakequant_bits(int bf16_dtype, const void* x, void* y, int64_t rows, int cols, float sig_max, float sig_min, int bits, int flags,
                             int n_cu, hipStream_t stream);   // fq_quant.hip
int fq_launch_kron64_multi(int bf16_dtype, const void* jobs, int n_jobs, int bpj, const FqQuantOut& out, hipStream_t stream);
int fq_launch_rowmm(int bf16_dtype, const void* x, const void* Tm, void* y, int64_t rows, int n, int n_cu, hipStream_t stream);
int fq_launch_fwht_f32(const f16* x, float* y32, int64_t vecs, int P, float scale, int n_cu, hipStream_t stream);
int fq_launch_had_mfma(const f16* x, int64_t rows, int n, int K, const f16* hadK, float scale, float sig_max, float sig_min,
                       uint8_t* q_out, f16* scale_out, f16* y_out, int n_cu, hipStream_t stream, const f16* up = nullptr,
                       int plain_quantizer = 0);
int fq_launch_hadamard_quant(const f16* x, int64_t rows, int n, int K, const f16* hadK, float scale, float sig_max,
                             float sig_min, uint8_t* q, f16* scale_out, int n_cu, hipStream_t stream);
int fq_launch_gemm_i4(const uint8_t* X, const uint8_t* W, int64_t M, int N, int K, int32_t* c, f16* y, const f16* srow,
                      const f16* scol, const f16* bias, hipStream_t stream);
int fq_launch_rmsnorm(const f16* x, f16* y, int64_t rows, int cols, float eps, int n_cu, hipStream_t stream);
int fq_launch_kv_quant(const f16* x, const f16* T, int64_t rows, int hd, float cmax, float cmin, bool lac, uint8_t* q,
                       f16* param, f16* y, int n_cu, hipStream_t stream);
int fq_launch_kv_dequant(const uint8_t* q, const f16* param, int64_t rows, int hd, bool lac, f16* y, int n_cu,
                         hipStream_t stream);
int fq_launch_kv_quant_append(const f16* k, const f16* v, const f16* T, int64_t tokens, int src_heads, int hd, const float* clip4,
                              bool lac, void* kv_data, void* kv_param, const int* indptr, const int* indices, const int* last,
                              int num_layers, int layer_idx, int num_heads, int page_size, int added, int group, int n_cu,
                              hipStream_t stream);
int64_t fq_i4_frag_bytes(int N, int K);
int fq_launch_i4_to_frag(const uint8_t* W, int N, int K, void* img, int n_cu, hipStream_t stream);
int fq_launch_gemm_i4_skinny_multi(int n, const uint8_t* const* X, const void* const* wimg, int64_t M, const int* N, int K, f16* const* y,
                                   const f16* const* srow, const f16* const* scol, const f16* const* bias, hipStream_t stream);
int64_t fq_gemm_i4_skinny_split_ws_bytes(int64_t M, int N, int K);
int fq_launch_gemm_i4_skinny_split(const uint8_t* X, const void* wimg, int64_t M, int N, int K, f16* y, const f16* srow, const f16* scol,
                                   const f16* bias, int* kws, hipStream_t stream);
int fq_launch_gemm_i4_skinny(const uint8_t* X, const void* wimg, int64_t M, int N, int K, int32_t* c, f16* y, const f16* srow,
                             const f16* scol, const f16* bias, hipStream_t stream);
int64_t fq_bf6_blob_bytes(int64_t rows, int K);  // fq_gemm_bf6.hip (exported as is)
int fq_launch_i4_to_bf6(const uint8_t* q, int64_t rows, int K, int perm, uint8_t* blob, int n_cu, hipStream_t stream);
int fq_launch_i4_to_bf6_multi(int nsrc, const uint8_t* const* q, int64_t rows, int K, int perm, uint8_t* blob, int n_cu, hipStream_t stream);
int fq_launch_gemm_bf6(const uint8_t* xblob, const uint8_t* wblob, int64_t M, int N, int K, int32_t* c, f16* y,
                       const f16* srow, const f16* scol, const f16* bias, hipStream_t stream);
int fq_launch_kv_append(void* kv_data, void* kv_param, const int* indptr, const int* indices, const int* last, const uint8_t* k,
                        const uint8_t* v, const f16* kparam, const f16* vparam, const int* seqlen_indptr, int64_t total_tokens,
                        int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, int batch, int group, int n_cu,
                        hipStream_t stream, bool f16_cache = false);
int fq_launch_kv_decode(f16* o, const f16* q, void* kv_data, void* kv_param, const int* indptr, const int* indices, const int* last,
                        int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, int batch, const f16* qt,
                        int transpose_out, hipStream_t stream, bool f16_cache = false, float* ws = nullptr, int splits = 1, int qgroup = 1,
                        const f16* k_new = nullptr, const f16* v_new = nullptr, const void* t_image = nullptr, int src_heads = 1, int copies = 1);
int64_t fq_kv_timage_bytes(int hd);
int fq_launch_kv_timage(const f16* T, int hd, void* img, hipStream_t stream);
int fq_kv_decode_splits(int batch, int num_heads, int seq_hint);
int fq_kv_decode_wg_heads(int batch, int num_q_heads, int q_group, int head_dim);
int64_t fq_kv_decode_ws_bytes_gqa(int batch, int num_q_heads, int q_group, int head_dim);
int64_t fq_kv_decode_ws_bytes(int batch, int num_heads, int head_dim);
int fq_launch_silu_mul(const f16* gate, const f16* up, f16* y, int64_t n, int n_cu, hipStream_t stream);
int fq_launch_silu_hadamard_quant(const f16* gate, const f16* up, int64_t rows, int n, int K, const f16* hadK, float scale,
                                  float sig_max, float sig_min, uint8_t* q, f16* scale_out, int n_cu, hipStream_t stream);
int64_t fq_kron_generic_workspace_bytes(int M, int N);
int fq_launch_kron_prepare(const f16* left, const f16* right, int M, int N, void* workspace, hipStream_t stream, int groups = 1);
int fq_launch_kron_grouped_mats(int flags, const void* x, const void* left, const void* right, int64_t rows, int M, int N,
                                const FqQuantOut& out, int n_groups, void* workspace, int64_t workspace_bytes, int n_cu,
                                hipStream_t stream);
int fq_launch_block(int flags, const f16* x, const f16* P, int64_t rows, int R, int C, int transpose_out,
                    const FqQuantOut& out, int n_cu, hipStream_t stream);
int fq_launch_hadamard(const f16* x, f16* y, int64_t rows, int n, int K, const f16* hadK, float scale,
                       int n_cu, hipStream_t stream);
int fq_launch_rowquant(int flags, const f16* x, int64_t rows, int cols, const FqQuantOut& out, int n_cu,
                       hipStream_t stream);
int fq_launch_sym_quant(const f16* x, const f16* scale, int64_t rows, int cols, uint8_t* q, int n_cu,
                        hipStream_t stream);
int fq_launch_sym_dequant(const int32_t* q, const f16* srow, const f16* scol, int64_t rows, int cols,
                          f16* x, int n_cu, hipStream_t stream);
int fq_launch_kron64_trace(const f16* x, const f16* left, const f16* right, int64_t rows, const FqQuantOut& out,
                           unsigned long long* trace, int n_cu, hipStream_t stream);

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// Every tensor the kernels touch with 16-byte accesses (activations, factor matrices, packed / fp16 outputs, workspaces, cache
// pages) must start on a 16-byte boundary: torch allocations do (256 B), views that start mid-row may not. NULL passes (optional).
bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }
#define FQ_NEED_ALIGN16(what, ...)                                                                                   \
    do {                                                                                                             \
        const void* fq_ptrs_[] = {__VA_ARGS__};                                                                      \
        for (size_t fq_i_ = 0; fq_i_ < sizeof(fq_ptrs_) / sizeof(fq_ptrs_[0]); ++fq_i_)                              \
            if (misaligned16(fq_ptrs_[fq_i_]))                                                                       \
                return fail(FQ_EINVAL, "%s: pointer argument %d of {" #__VA_ARGS__ "} is not 16-byte aligned", what, (int)fq_i_); \
    } while (0)

int cu_count() {
    (void)hipGetLastError();  // drop any stale error of this thread so the post-launch check is ours
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    static int cached[64] = {0};  // immutable per device once written; a racy double-write stores the same value
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

int check_launch(int rc, const char* what) {
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "%s: flag combination / shape has no compiled kernel", what);
    if (rc != 0) return fail(FQ_ELAUNCH, "%s: HIP launch failed: %s", what, hipGetErrorString((hipError_t)rc));
    return FQ_OK;
}

// Fill FqQuantOut from the C-ABI arguments; validates the pointers the flags require.
int fill_out(const char* what, FqQuantOut& o, const float* sig_max, const float* sig_min, int n_clips,
             int flags, void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out) {
    memset(&o, 0, sizeof(o));
    const int outs = flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM);
    if (outs == 0) return fail(FQ_EINVAL, "%s: flags select no output", what);
    if (flags & ~(FQ_OUT_PACKED | FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM | FQ_ROUND_Y_F16 | FQ_NO_CLAMP0 | FQ_WS_PREPARED |
                  FQ_QUANT_F16 | FQ_GROUP128 | FQ_SIG_F16 | FQ_RATIO_POST))
        return fail(FQ_EINVAL, "%s: unknown flag bits 0x%x", what, flags);
    if (flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT)) {
        if (n_clips < 1 || n_clips > FQ_MAX_CLIPS)
            return fail(FQ_EINVAL, "%s: n_clips=%d out of [1,%d]", what, n_clips, FQ_MAX_CLIPS);
        if (!sig_max || !sig_min) return fail(FQ_EINVAL, "%s: sig_max/sig_min is NULL", what);
        o.n_clips = n_clips;
        for (int i = 0; i < n_clips; ++i) {
            o.sig_max[i] = sig_max[i];
            o.sig_min[i] = sig_min[i];
            if (flags & FQ_OUT_PACKED) {
                if (!q_out || !scale_out || !q_out[i] || !scale_out[i])
                    return fail(FQ_EINVAL, "%s: FQ_OUT_PACKED needs q_out[%d] and scale_out[%d]", what, i, i);
                o.q[i] = (uint8_t*)q_out[i];
                o.scale[i] = (f16*)scale_out[i];
            }
            if (flags & FQ_OUT_FAKEQUANT) {
                if (!fq_out || !fq_out[i]) return fail(FQ_EINVAL, "%s: FQ_OUT_FAKEQUANT needs fq_out[%d]", what, i);
                o.fq[i] = (f16*)fq_out[i];
            }
        }
    }
    if (flags & FQ_OUT_TRANSFORM) {
        if (!y_out) return fail(FQ_EINVAL, "%s: FQ_OUT_TRANSFORM needs y_out", what);
        o.y = (f16*)y_out;
    }
    for (int i = 0; i < FQ_MAX_CLIPS; ++i) FQ_NEED_ALIGN16(what, o.q[i], o.fq[i]);
    FQ_NEED_ALIGN16(what, o.y);
    o.rt_flags = flags & (FQ_ROUND_Y_F16 | FQ_NO_CLAMP0 | FQ_GROUP128 | FQ_SIG_F16 | FQ_RATIO_POST);
    o.rms_eps = 0.0f;
    o.in2 = nullptr;
    return FQ_OK;
}

}  // namespace

// Kernel selection shared by fq_kron_quant_f16 and fq_kron_quant_grouped_f16 (o carries the outputs, the run-time flags
// and, for a grouped launch, the group arrays).
static int kron_dispatch(const char* what, const FqQuantOut& o, int flags, const void* x, const void* left, const void* right,
                         const void* diag, int64_t rows, int M, int N, void* workspace, int64_t workspace_bytes,
                         void* stream, int dt = 0) {
    int rc;
    flags |= dt;  // FQ_DT_BF16 (internal): the launchers pick the bf16 instantiations
    FQ_NEED_ALIGN16(what, x, left, right, diag, workspace);
    const int n_cu = cu_count();
    const bool special = o.group_offsets != nullptr || (o.rt_flags & FQ_GROUP128);  // only the fused MFMA kernels take these
    // FQ_GROUP128: packed-only fp32-arithmetic launches at N = 64 quantise the fp32 accumulator (wave-per-token kernels); every
    // other output set / arithmetic / pair runs the group epilogue of the workgroup-per-token kernel, which quantises the
    // transformed activation ROUNDED to the activation dtype — what ActivationQuantizer(groupsize=128) is handed (path A)
    if ((o.rt_flags & FQ_GROUP128) && ((M * N) % 128 != 0 || o.n_clips != 1 ||
                                       ((flags & (FQ_QUANT_F16 | FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM)) && !(flags & FQ_ROUND_Y_F16))))
        return fail(FQ_EUNSUPPORTED, "%s: FQ_GROUP128 needs one clip set, M*N %% 128 == 0, and FQ_ROUND_Y_F16 for any output set other "
                    "than packed with fp32 arithmetic", what);
    if (M == 64 && N == 64) {
        // the workspace is OPTIONAL at 64 x 64: with one (>= 16 KB) the kernel reads the fragment image from it (written here
        // unless FQ_WS_PREPARED says it is there already), without one it gathers the fragments from the matrices itself
        const void* prep = (workspace && workspace_bytes >= FQ_K64_WS_BYTES) ? workspace : nullptr;
        if (prep && !(flags & FQ_WS_PREPARED)) {
            rc = fq_launch_kron64_prepare(left, right, workspace, (hipStream_t)stream);
            if (rc == 0) rc = fq_launch_kron_prepare((const f16*)left, (const f16*)right, 64, 64,
                                                     static_cast<unsigned char*>(workspace) + FQ_K64_IMAGE_BYTES, (hipStream_t)stream);
            if (rc != 0) return check_launch(rc, what);
            flags |= FQ_WS_PREPARED;   // (both images are there now, also for the fall-through below)
        }
        rc = fq_launch_kron64(flags & ~FQ_WS_PREPARED, (const f16*)x, (const f16*)left, (const f16*)right, (const f16*)diag,
                              rows, o, n_cu, (hipStream_t)stream, prep);
        if (rc != -1000) return check_launch(rc, what);
        if (dt) return fail(FQ_EUNSUPPORTED, "%s: output set 0x%x has no bf16 kernel at 64 x 64", what, flags & ~dt);
        if ((o.rt_flags & FQ_GROUP128) && !(flags & FQ_ROUND_Y_F16))
            return fail(FQ_EUNSUPPORTED, "%s: FQ_GROUP128 at 64 x 64 needs the packed-only output set or FQ_ROUND_Y_F16", what);
        if (prep) {   // the second half of the 64 x 64 workspace is the other kernel family's image
            workspace = static_cast<unsigned char*>(workspace) + FQ_K64_IMAGE_BYTES;
            workspace_bytes -= FQ_K64_IMAGE_BYTES;
        } else {
            flags &= ~FQ_WS_PREPARED;
        }
    }
    rc = fq_launch_kron_generic(flags, (const f16*)x, (const f16*)left, (const f16*)right, (const f16*)diag,
                                rows, M, N, o, workspace, workspace_bytes, n_cu, (hipStream_t)stream);
    if (rc == -1001)
        return fail(FQ_EINVAL, "%s: workspace of %lld bytes required for M=%d N=%d (got %lld)", what,
                    (long long)fq_kron_generic_workspace_bytes(M, N), M, N, (long long)(workspace ? workspace_bytes : 0));
    if (rc == -1000 && special)
        return fail(FQ_EUNSUPPORTED, "%s: grouped / FQ_GROUP128 launches need a fused MFMA kernel; factors (%d, %d) with "
                    "flags 0x%x have none", what, M, N, flags);
    if (rc == -1000)
        return fail(FQ_EUNSUPPORTED, "%s: no kernel for factors (%d, %d) with flags 0x%x: M, N <= 256, N even, M*N <= 32768", what, M, N, flags);
    return check_launch(rc, what);
}

extern "C" {

const char* fq_last_error(void) { return g_err; }
int fq_version(void) { return 120; }

static int kron_quant_impl(const char* what, int dt, const void* x, const void* left, const void* right, const void* diag, int64_t rows,
                           int M, int N, const float* sig_max, const float* sig_min, int n_clips, int flags,
                           void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                           void* workspace, int64_t workspace_bytes, void* stream) {
    if (rows < 0 || M <= 0 || N <= 0) return fail(FQ_EINVAL, "%s: bad sizes rows=%lld M=%d N=%d", what, (long long)rows, M, N);
    if (N & 1) return fail(FQ_EINVAL, "%s: N=%d must be even (two INT4 per byte)", what, N);
    if (flags & FQ_RATIO_POST) return fail(FQ_EINVAL, "%s: FQ_RATIO_POST is a row-quantiser flag", what);
    FqQuantOut o;
    int rc = fill_out(what, o, sig_max, sig_min, n_clips, flags, q_out, scale_out, fq_out, y_out);
    if (rc != FQ_OK) return rc;
    if (rows == 0) return FQ_OK;  // empty batch: nothing to launch (zero-size tensors have NULL data)
    if (!x || !left || !right) return fail(FQ_EINVAL, "%s: x/left/right is NULL", what);
    return kron_dispatch(what, o, flags, x, left, right, diag, rows, M, N, workspace, workspace_bytes, stream, dt);
}

int fq_kron_quant_f16(const void* x, const void* left, const void* right, const void* diag, int64_t rows,
                      int M, int N, const float* sig_max, const float* sig_min, int n_clips, int flags,
                      void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                      void* workspace, int64_t workspace_bytes, void* stream) {
    return kron_quant_impl("fq_kron_quant_f16", 0, x, left, right, diag, rows, M, N, sig_max, sig_min, n_clips, flags, q_out,
                           scale_out, fq_out, y_out, workspace, workspace_bytes, stream);
}

int fq_kron_quant_bf16(const void* x, const void* left, const void* right, const void* diag, int64_t rows,
                       int M, int N, const float* sig_max, const float* sig_min, int n_clips, int flags,
                       void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                       void* workspace, int64_t workspace_bytes, void* stream) {
    return kron_quant_impl("fq_kron_quant_bf16", FQ_DT_BF16, x, left, right, diag, rows, M, N, sig_max, sig_min, n_clips, flags,
                           q_out, scale_out, fq_out, y_out, workspace, workspace_bytes, stream);
}

static int kron_quant_grouped_impl(const char* what, int dt, const void* x, const void* left, const void* right, int64_t rows, int M, int N,
                              const int64_t* group_offsets, int n_groups, const float* sig_max_g, const float* sig_min_g,
                              int flags, void* q_out, void* scale_out, void* fq_out, void* y_out,
                              void* workspace, int64_t workspace_bytes, void* stream) {
    if (rows < 0 || M <= 0 || N <= 0) return fail(FQ_EINVAL, "%s: bad sizes rows=%lld M=%d N=%d", what, (long long)rows, M, N);
    if (N & 1) return fail(FQ_EINVAL, "%s: N=%d must be even (two INT4 per byte)", what, N);
    if (n_groups < 1) return fail(FQ_EINVAL, "%s: n_groups=%d", what, n_groups);
    if ((flags & FQ_QUANT_F16) && (flags & FQ_OUT_PACKED))
        return fail(FQ_EUNSUPPORTED, "%s: the low-precision quantiser arithmetic is offered for the fake-quant output only", what);
    const bool quant = (flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT)) != 0;
    const float one = 1.0f;  // fill_out wants a host clip pair; the kernels read the per-group device arrays instead
    void* q1[FQ_MAX_CLIPS] = {q_out}, *s1[FQ_MAX_CLIPS] = {scale_out}, *f1[FQ_MAX_CLIPS] = {fq_out};
    FqQuantOut o;
    int rc = fill_out(what, o, &one, &one, 1, flags, q1, s1, f1, y_out);
    if (rc != FQ_OK) return rc;
    if (rows == 0) return FQ_OK;
    if (!x || !left || !right) return fail(FQ_EINVAL, "%s: x/left/right is NULL", what);
    if (!group_offsets || (quant && (!sig_max_g || !sig_min_g)))
        return fail(FQ_EINVAL, "%s: group_offsets / sig_max_g / sig_min_g is NULL", what);
    if (quant) {  // a transform-only launch needs no clip pairs: leave it ungrouped
        o.group_offsets = group_offsets;
        o.sig_max_g = sig_max_g;
        o.sig_min_g = sig_min_g;
        o.n_groups = n_groups;
    }
    return kron_dispatch(what, o, flags, x, left, right, nullptr, rows, M, N, workspace, workspace_bytes, stream, dt);
}

int fq_kron_quant_grouped_f16(const void* x, const void* left, const void* right, int64_t rows, int M, int N,
                              const int64_t* group_offsets, int n_groups, const float* sig_max_g, const float* sig_min_g,
                              int flags, void* q_out, void* scale_out, void* fq_out, void* y_out,
                              void* workspace, int64_t workspace_bytes, void* stream) {
    return kron_quant_grouped_impl("fq_kron_quant_grouped_f16", 0, x, left, right, rows, M, N, group_offsets, n_groups, sig_max_g,
                                   sig_min_g, flags, q_out, scale_out, fq_out, y_out, workspace, workspace_bytes, stream);
}

int fq_kron_quant_grouped_bf16(const void* x, const void* left, const void* right, int64_t rows, int M, int N,
                               const int64_t* group_offsets, int n_groups, const float* sig_max_g, const float* sig_min_g,
                               int flags, void* q_out, void* scale_out, void* fq_out, void* y_out,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    return kron_quant_grouped_impl("fq_kron_quant_grouped_bf16", FQ_DT_BF16, x, left, right, rows, M, N, group_offsets, n_groups,
                                   sig_max_g, sig_min_g, flags, q_out, scale_out, fq_out, y_out, workspace, workspace_bytes, stream);
}

static int kron_quant_grouped_mats_impl(const char* what, int dt, const void* x, const void* left_g, const void* right_g, int64_t rows,
                                        int M, int N, const int64_t* group_offsets, int n_groups, const float* sig_max_g,
                                        const float* sig_min_g, int flags, void* q_out, void* scale_out, void* fq_out, void* y_out,
                                        void* workspace, int64_t workspace_bytes, void* stream) {
    if (rows < 0 || M <= 0 || N <= 0) return fail(FQ_EINVAL, "%s: bad sizes rows=%lld M=%d N=%d", what, (long long)rows, M, N);
    if (N & 1) return fail(FQ_EINVAL, "%s: N=%d must be even (two INT4 per byte)", what, N);
    if (n_groups < 1) return fail(FQ_EINVAL, "%s: n_groups=%d", what, n_groups);
    if (flags & (FQ_GROUP128 | FQ_RATIO_POST)) return fail(FQ_EUNSUPPORTED, "%s: per-token scales only", what);
    const bool quant = (flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT)) != 0;
    const float one = 1.0f;
    void* q1[FQ_MAX_CLIPS] = {q_out}, *s1[FQ_MAX_CLIPS] = {scale_out}, *f1[FQ_MAX_CLIPS] = {fq_out};
    FqQuantOut o;
    int rc = fill_out(what, o, &one, &one, 1, flags, q1, s1, f1, y_out);
    if (rc != FQ_OK) return rc;
    if (rows == 0) return FQ_OK;
    if (!x || !left_g || !right_g || !group_offsets) return fail(FQ_EINVAL, "%s: x / left_g / right_g / group_offsets is NULL", what);
    if (quant && (!sig_max_g || !sig_min_g)) return fail(FQ_EINVAL, "%s: sig_max_g / sig_min_g is NULL", what);
    FQ_NEED_ALIGN16(what, x, left_g, right_g, workspace);
    if ((((int64_t)M * M * 2) & 15) || (((int64_t)N * N * 2) & 15))
        return fail(FQ_EUNSUPPORTED, "%s: the per-group matrices must each start on a 16-byte boundary (M*M and N*N multiples of 8)", what);
    o.group_offsets = group_offsets;     // (the matrices follow the groups even in a transform-only launch)
    o.sig_max_g = quant ? sig_max_g : nullptr;
    o.sig_min_g = quant ? sig_min_g : nullptr;
    o.n_groups = n_groups;
    rc = fq_launch_kron_grouped_mats(flags | dt, x, left_g, right_g, rows, M, N, o, n_groups, workspace, workspace_bytes, cu_count(),
                                     (hipStream_t)stream);
    if (rc == -1001)
        return fail(FQ_EINVAL, "%s: workspace of %lld bytes required for %d groups of M=%d N=%d (got %lld)", what,
                    (long long)(fq_kron_generic_workspace_bytes(M, N) * n_groups), n_groups, M, N, (long long)(workspace ? workspace_bytes : 0));
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "%s: no workgroup-per-token kernel for factors (%d, %d)", what, M, N);
    return check_launch(rc, what);
}

int fq_kron_quant_grouped_mats_f16(const void* x, const void* left_g, const void* right_g, int64_t rows, int M, int N,
                                   const int64_t* group_offsets, int n_groups, const float* sig_max_g, const float* sig_min_g,
                                   int flags, void* q_out, void* scale_out, void* fq_out, void* y_out,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    return kron_quant_grouped_mats_impl("fq_kron_quant_grouped_mats_f16", 0, x, left_g, right_g, rows, M, N, group_offsets, n_groups,
                                        sig_max_g, sig_min_g, flags, q_out, scale_out, fq_out, y_out, workspace, workspace_bytes, stream);
}

int fq_kron_quant_grouped_mats_bf16(const void* x, const void* left_g, const void* right_g, int64_t rows, int M, int N,
                                    const int64_t* group_offsets, int n_groups, const float* sig_max_g, const float* sig_min_g,
                                    int flags, void* q_out, void* scale_out, void* fq_out, void* y_out,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
    return kron_quant_grouped_mats_impl("fq_kron_quant_grouped_mats_bf16", FQ_DT_BF16, x, left_g, right_g, rows, M, N, group_offsets,
                                        n_groups, sig_max_g, sig_min_g, flags, q_out, scale_out, fq_out, y_out, workspace, workspace_bytes,
                                        stream);
}

int fq_rmsnorm_kron_quant_ws_f16(const void* x, float eps, const void* left, const void* right, int64_t rows, int M, int N,
                                 const float* sig_max, const float* sig_min, int n_clips, int flags,
                                 void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
    const char* what = "fq_rmsnorm_kron_quant_ws_f16";
    if (M == 64 && N == 64)   // (the 64 x 64 kernel gathers its fragments itself when the launch carries no image)
        return fq_rmsnorm_kron_quant_f16(x, eps, left, right, rows, M, N, sig_max, sig_min, n_clips, flags & ~FQ_WS_PREPARED, q_out,
                                         scale_out, fq_out, y_out, stream);
    if (rows < 0 || M <= 0 || N <= 0 || (N & 1)) return fail(FQ_EINVAL, "%s: bad sizes", what);
    if (!(eps >= 0.0f)) return fail(FQ_EINVAL, "%s: eps must be >= 0", what);
    if ((flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM | FQ_QUANT_F16)) != FQ_OUT_PACKED || (flags & (FQ_GROUP128 | FQ_RATIO_POST)))
        return fail(FQ_EUNSUPPORTED, "%s: packed output only for pairs other than 64 x 64", what);
    FqQuantOut o;
    int rc = fill_out(what, o, sig_max, sig_min, n_clips, flags, q_out, scale_out, fq_out, y_out);
    if (rc != FQ_OK) return rc;
    if (rows == 0) return FQ_OK;
    if (!x || !left || !right) return fail(FQ_EINVAL, "%s: x/left/right is NULL", what);
    FQ_NEED_ALIGN16(what, x, left, right, workspace);
    o.rms_eps = eps;
    rc = fq_launch_kron_generic(flags | FQ_IN_RMSNORM, (const f16*)x, (const f16*)left, (const f16*)right, nullptr, rows, M, N, o,
                                workspace, workspace_bytes, cu_count(), (hipStream_t)stream);
    if (rc == -1001)
        return fail(FQ_EINVAL, "%s: workspace of %lld bytes required for M=%d N=%d (got %lld)", what,
                    (long long)fq_kron_generic_workspace_bytes(M, N), M, N, (long long)(workspace ? workspace_bytes : 0));
    if (rc == -1000)
        return fail(FQ_EUNSUPPORTED, "%s: the RMSNorm is fused for 64 x 64 and the wave-per-token pairs (M <= 64, N in {64, 80, 112, 128}); "
                    "run fq_rmsnorm_f16 first for (%d, %d)", what, M, N);
    return check_launch(rc, what);
}

int fq_rmsnorm_kron_quant_f16(const void* x, float eps, const void* left, const void* right, int64_t rows, int M, int N,
                              const float* sig_max, const float* sig_min, int n_clips, int flags,
                              void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                              void* stream) {
    if (rows < 0 || M <= 0 || N <= 0) return fail(FQ_EINVAL, "fq_rmsnorm_kron_quant_f16: bad sizes");
    if (!(eps >= 0.0f)) return fail(FQ_EINVAL, "fq_rmsnorm_kron_quant_f16: eps must be >= 0");
    if (M != 64 || N != 64) return fail(FQ_EUNSUPPORTED, "fq_rmsnorm_kron_quant_f16: only 64 x 64 factors are fused (got %d x %d)", M, N);
    if (flags & (FQ_QUANT_F16 | FQ_OUT_FAKEQUANT)) return fail(FQ_EUNSUPPORTED, "fq_rmsnorm_kron_quant_f16: packed / transform outputs only");
    FqQuantOut o;
    int rc = fill_out("fq_rmsnorm_kron_quant_f16", o, sig_max, sig_min, n_clips, flags, q_out, scale_out, fq_out, y_out);
    if (rc != FQ_OK) return rc;
    if (rows == 0) return FQ_OK;
    if (!x || !left || !right) return fail(FQ_EINVAL, "fq_rmsnorm_kron_quant_f16: x/left/right is NULL");
    FQ_NEED_ALIGN16("fq_rmsnorm_kron_quant_f16", x, left, right);
    o.rms_eps = eps;
    rc = fq_launch_kron64(flags | FQ_IN_RMSNORM, (const f16*)x, (const f16*)left, (const f16*)right, nullptr, rows, o,
                          cu_count(), (hipStream_t)stream);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "fq_rmsnorm_kron_quant_f16: output set 0x%x has no fused kernel", flags);
    return check_launch(rc, "fq_rmsnorm_kron_quant_f16");
}

int fq_silu_mul_kron_quant_f16(const void* gate, const void* up, const void* left, const void* right, int64_t rows,
                               int M, int N, const float* sig_max, const float* sig_min, int n_clips, int flags,
                               void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    if (rows < 0 || M <= 0 || N <= 0 || (N & 1)) return fail(FQ_EINVAL, "fq_silu_mul_kron_quant_f16: bad sizes rows=%lld M=%d N=%d", (long long)rows, M, N);
    FqQuantOut o;
    int rc = fill_out("fq_silu_mul_kron_quant_f16", o, sig_max, sig_min, n_clips, flags, q_out, scale_out, fq_out, y_out);
    if (rc != FQ_OK) return rc;
    if (rows == 0) return FQ_OK;
    if (!gate || !up || !left || !right) return fail(FQ_EINVAL, "fq_silu_mul_kron_quant_f16: gate/up/left/right is NULL");
    FQ_NEED_ALIGN16("fq_silu_mul_kron_quant_f16", gate, up, left, right, workspace);
    o.in2 = (const f16*)up;
    rc = fq_launch_kron_generic(flags | FQ_IN_SILU_MUL, (const f16*)gate, (const f16*)left, (const f16*)right, nullptr,
                                rows, M, N, o, workspace, workspace_bytes, cu_count(), (hipStream_t)stream);
    if (rc == -1001)
        return fail(FQ_EINVAL, "fq_silu_mul_kron_quant_f16: workspace of %lld bytes required for M=%d N=%d (got %lld)",
                    (long long)fq_kron_generic_workspace_bytes(M, N), M, N, (long long)(workspace ? workspace_bytes : 0));
    if (rc == -1000)
        return fail(FQ_EUNSUPPORTED, "fq_silu_mul_kron_quant_f16: no fused kernel for M=%d N=%d (use fq_silu_mul_f16 + fq_kron_quant_f16)", M, N);
    return check_launch(rc, "fq_silu_mul_kron_quant_f16");
}

int fq_kron_quant_ex_f16(const void* x, const void* up, const void* left, const void* right, int64_t rows, int M, int N,
                         float post_scale, const float* sig_max, const float* sig_min, int n_clips, int flags,
                         void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                         void* workspace, int64_t workspace_bytes, void* stream) {
    const char* what = "fq_kron_quant_ex_f16";
    if (rows < 0 || M <= 0 || N <= 0 || (N & 1)) return fail(FQ_EINVAL, "%s: bad sizes rows=%lld M=%d N=%d", what, (long long)rows, M, N);
    if (!(post_scale > 0.0f) || !(post_scale < 3.0e38f)) return fail(FQ_EINVAL, "%s: post_scale must be a positive finite number", what);
    FqQuantOut o;
    int rc = fill_out(what, o, sig_max, sig_min, n_clips, flags, q_out, scale_out, fq_out, y_out);
    if (rc != FQ_OK) return rc;
    if (rows == 0) return FQ_OK;
    if (!x || !left || !right) return fail(FQ_EINVAL, "%s: x/left/right is NULL", what);
    FQ_NEED_ALIGN16(what, x, up, left, right, workspace);
    // FQ_RATIO_POST (deploy.nn.Quantizer(lac=False) behind the rotation: scale = fp16(max|y| / 7) * sig_max, no zero guard): the tall kernel
    // (64 < M <= 192, N = 64: 11008 = 172 x 64, ...) is the one Kronecker kernel that stores the reference's scale for an all-zero token
    if ((flags & FQ_RATIO_POST) && !(N == 64 && M > 64 && M <= 192 && !up &&
                                     (flags & ((FQ_CT_MASK & ~FQ_OUT_TRANSFORM) | FQ_ROUND_Y_F16 | FQ_SIG_F16)) ==
                                         (FQ_OUT_PACKED | FQ_QUANT_F16 | FQ_ROUND_Y_F16)))
        return fail(FQ_EUNSUPPORTED, "%s: FQ_RATIO_POST goes with FQ_OUT_PACKED [| FQ_OUT_TRANSFORM] | FQ_QUANT_F16 | FQ_ROUND_Y_F16 on 64 < M <= 192, N = 64 (no up)", what);
    o.post_scale = post_scale == 1.0f ? 0.0f : post_scale;
    o.in2 = (const f16*)up;
    // the scaled / SiLU.mul forms live in the workgroup-per-token kernel family only (the shapes this entry exists for:
    // M > 64, e.g. a Hadamard rotation of n = K * P as the Kronecker product (hadK x H_{P/N}) (x) H_N)
    rc = fq_launch_kron_generic(flags | (up ? FQ_IN_SILU_MUL : 0) | FQ_NO_WAVE_KERNEL, (const f16*)x, (const f16*)left, (const f16*)right,
                                nullptr, rows, M, N, o, workspace, workspace_bytes, cu_count(), (hipStream_t)stream);
    if (rc == -1001)
        return fail(FQ_EINVAL, "%s: workspace of %lld bytes required for M=%d N=%d (got %lld)", what,
                    (long long)fq_kron_generic_workspace_bytes(M, N), M, N, (long long)(workspace ? workspace_bytes : 0));
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "%s: no kernel for M=%d N=%d with flags 0x%x", what, M, N, flags);
    return check_launch(rc, what);
}

int fq_silu_mul_f16(const void* gate, const void* up, void* y, int64_t n, void* stream) {
    FQ_NEED_ALIGN16("fq_silu_mul_f16", gate, up, y);
    if (n < 0) return fail(FQ_EINVAL, "fq_silu_mul_f16: n < 0");
    if (n == 0) return FQ_OK;
    if (!gate || !up || !y) return fail(FQ_EINVAL, "fq_silu_mul_f16: NULL pointer");
    const int rc = fq_launch_silu_mul((const f16*)gate, (const f16*)up, (f16*)y, n, cu_count(), (hipStream_t)stream);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "fq_silu_mul_f16: n=%lld must be a multiple of 8", (long long)n);
    return check_launch(rc, "fq_silu_mul_f16");
}

int fq_silu_mul_hadamard_quant_f16(const void* gate, const void* up, int64_t rows, int n, int K, const void* hadK,
                                   float scale, float sig_max, float sig_min, void* q_out, void* scale_out,
                                   void* stream) {
    if (!q_out || !scale_out) return fail(FQ_EINVAL, "fq_silu_mul_hadamard_quant_f16: NULL pointer");
    if (rows < 0 || n <= 0 || K <= 0 || n % K) return fail(FQ_EINVAL, "fq_silu_mul_hadamard_quant_f16: bad sizes n=%d K=%d", n, K);
    if (K > 1 && !hadK) return fail(FQ_EINVAL, "fq_silu_mul_hadamard_quant_f16: hadK is NULL with K=%d", K);
    if (!(sig_max > 0.0f) || !(sig_min > 0.0f)) return fail(FQ_EINVAL, "fq_silu_mul_hadamard_quant_f16: sig_max/sig_min must be > 0");
    if (rows == 0) return FQ_OK;
    if (!gate || !up) return fail(FQ_EINVAL, "fq_silu_mul_hadamard_quant_f16: gate/up is NULL");
    FQ_NEED_ALIGN16("fq_silu_mul_hadamard_quant_f16", gate, up, hadK, q_out);
    const int rc = fq_launch_silu_hadamard_quant((const f16*)gate, (const f16*)up, rows, n, K, (const f16*)hadK, scale,
                                                 sig_max, sig_min, (uint8_t*)q_out, (f16*)scale_out, cu_count(),
                                                 (hipStream_t)stream);
    if (rc == -1000)
        return fail(FQ_EUNSUPPORTED, "fq_silu_mul_hadamard_quant_f16: no fused kernel for n=%d K=%d", n, K);
    return check_launch(rc, "fq_silu_mul_hadamard_quant_f16");
}

int fq_rmsnorm_f16(const void* x, void* y, int64_t rows, int cols, float eps, void* stream) {
    FQ_NEED_ALIGN16("fq_rmsnorm_f16", x, y);
    if (!x || !y) return fail(FQ_EINVAL, "fq_rmsnorm_f16: x/y is NULL");
    if (rows < 0 || cols <= 0 || !(eps >= 0.0f)) return fail(FQ_EINVAL, "fq_rmsnorm_f16: bad arguments");
    if (rows == 0) return FQ_OK;
    const int rc = fq_launch_rmsnorm((const f16*)x, (f16*)y, rows, cols, eps, cu_count(), (hipStream_t)stream);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "fq_rmsnorm_f16: cols=%d must be a multiple of 8 and <= 16384", cols);
    return check_launch(rc, "fq_rmsnorm_f16");
}

int fq_kron_prepare_f16(const void* left, const void* right, int M, int N, void* workspace, int64_t workspace_bytes,
                        void* stream) {
    if (M == 64 && N == 64) {   // optional image: written when a workspace is given, a no-op otherwise (as before round 3)
        if (!workspace || workspace_bytes < FQ_K64_WS_BYTES) return FQ_OK;
        if (!left || !right) return fail(FQ_EINVAL, "fq_kron_prepare_f16: left/right is NULL");
        FQ_NEED_ALIGN16("fq_kron_prepare_f16", left, right, workspace);
        int rc = fq_launch_kron64_prepare(left, right, workspace, (hipStream_t)stream);
        if (rc == 0) rc = fq_launch_kron_prepare((const f16*)left, (const f16*)right, 64, 64,
                                                 static_cast<unsigned char*>(workspace) + FQ_K64_IMAGE_BYTES, (hipStream_t)stream);
        return check_launch(rc, "fq_kron_prepare_f16");
    }
    const int64_t need = fq_kron_workspace_bytes(M, N);
    if (need < 0) return fail(FQ_EUNSUPPORTED, "fq_kron_prepare_f16: no kernel for factors (%d, %d)", M, N);
    if (!left || !right) return fail(FQ_EINVAL, "fq_kron_prepare_f16: left/right is NULL");
    if (!workspace || workspace_bytes < need)
        return fail(FQ_EINVAL, "fq_kron_prepare_f16: workspace of %lld bytes required (got %lld)", (long long)need,
                    (long long)(workspace ? workspace_bytes : 0));
    return check_launch(fq_launch_kron_prepare((const f16*)left, (const f16*)right, M, N, workspace, (hipStream_t)stream),
                        "fq_kron_prepare_f16");
}

int fq_kron_prepare_bf16(const void* left, const void* right, int M, int N, void* workspace, int64_t workspace_bytes,
                         void* stream) {
    // the fragment image re-arranges 16-bit words and pads with zero bits: one kernel for both element types
    return fq_kron_prepare_f16(left, right, M, N, workspace, workspace_bytes, stream);
}

int64_t fq_kron_workspace_bytes(int M, int N) {
    if (M == 64 && N == 64) return FQ_K64_WS_BYTES;   // optional (NULL still works): the prepared fragment images
    if (M < 1 || N < 2 || (N & 1) || M > 256 || N > 256 || (int64_t)M * N > 32768) return FQ_EUNSUPPORTED;
    return fq_kron_generic_workspace_bytes(M, N);
}

// ---- multi-job launch (fq_kron64.hip, FQ_K64_MULTI) ----
struct Kron64JobDev {   // = FqKron64Job of fq_kron64.hip
    const void* x;
    const void* prep;
    void* q;
    void* scale;
    int64_t rows, tpb;
};
static int multi_wg_per_job(int n_jobs) {
    const int n_cu = cu_count();
    return n_jobs >= n_cu ? 1 : (n_cu + n_jobs - 1) / n_jobs;   // ~ one persistent workgroup per CU over all jobs
}

int64_t fq_kron_multi_table_bytes(int n_jobs) { return n_jobs > 0 ? (int64_t)n_jobs * (int64_t)sizeof(Kron64JobDev) : 0; }

int fq_kron_multi_prepare(const FqKronJob* jobs, int n_jobs, void* table, int64_t table_bytes, void* stream) {
    const char* what = "fq_kron_multi_prepare";
    if (!jobs || n_jobs < 1 || n_jobs > 65536) return fail(FQ_EINVAL, "%s: jobs is NULL or n_jobs=%d out of [1, 65536]", what, n_jobs);
    if (!table || table_bytes < fq_kron_multi_table_bytes(n_jobs))
        return fail(FQ_EINVAL, "%s: table of %lld bytes required", what, (long long)fq_kron_multi_table_bytes(n_jobs));
    FQ_NEED_ALIGN16(what, table);
    const int bpj = multi_wg_per_job(n_jobs);
    Kron64JobDev* host = static_cast<Kron64JobDev*>(malloc((size_t)n_jobs * sizeof(Kron64JobDev)));
    if (!host) return fail(FQ_EINVAL, "%s: out of host memory", what);
    for (int j = 0; j < n_jobs; ++j) {
        const FqKronJob& jb = jobs[j];
        if (jb.rows < 0 || (jb.rows > 0 && (!jb.x || !jb.workspace || !jb.q || !jb.scale))) {
            free(host);
            return fail(FQ_EINVAL, "%s: job %d has a NULL pointer or rows < 0", what, j);
        }
        if (((uintptr_t)jb.x | (uintptr_t)jb.workspace | (uintptr_t)jb.q) & 15) {
            free(host);
            return fail(FQ_EINVAL, "%s: job %d: x / workspace / q must be 16-byte aligned", what, j);
        }
        host[j] = Kron64JobDev{jb.x, jb.workspace, jb.q, jb.scale, jb.rows, (jb.rows + bpj - 1) / bpj};
    }
    // a set-up call (once per model): the copy is waited for here, so `host` and the caller's array can go at once
    hipError_t e = hipMemcpyAsync(table, host, (size_t)n_jobs * sizeof(Kron64JobDev), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free(host);
    if (e != hipSuccess) return check_launch((int)e, what);
    return bpj;
}

static int kron_multi_impl(const char* what, int bf16_dtype, const void* table, int n_jobs, int wg_per_job, float sig_max, float sig_min,
                           int flags, void* stream) {
    if (!table || n_jobs < 1 || n_jobs > 65536 || wg_per_job < 1) return fail(FQ_EINVAL, "%s: table is NULL, n_jobs=%d or wg_per_job=%d", what, n_jobs, wg_per_job);
    if ((flags & ~(FQ_OUT_PACKED | FQ_NO_CLAMP0 | FQ_WS_PREPARED)) || !(flags & FQ_OUT_PACKED))
        return fail(FQ_EUNSUPPORTED, "%s: flags 0x%x (FQ_OUT_PACKED, optionally FQ_NO_CLAMP0)", what, flags);
    if (!(sig_max > 0.0f) || !(sig_min > 0.0f)) return fail(FQ_EINVAL, "%s: sig_max/sig_min must be > 0", what);
    if ((int64_t)n_jobs * wg_per_job > 0x7fffffff) return fail(FQ_EINVAL, "%s: grid too large", what);
    // the per-job tokens-per-workgroup in the table were baked by fq_kron_multi_prepare for ITS workgroup count (a function of n_jobs
    // and the device): with fewer workgroups per job than that the tail tokens of every job would silently never be written
    if (wg_per_job != multi_wg_per_job(n_jobs))
        return fail(FQ_EINVAL, "%s: wg_per_job=%d but fq_kron_multi_prepare(n_jobs=%d) returned %d for this table", what, wg_per_job, n_jobs,
                    multi_wg_per_job(n_jobs));
    (void)cu_count();   // (drops a stale error of this thread)
    FqQuantOut o;
    memset(&o, 0, sizeof(o));
    o.n_clips = 1;
    o.sig_max[0] = sig_max;
    o.sig_min[0] = sig_min;
    o.rt_flags = flags & FQ_NO_CLAMP0;
    return check_launch(fq_launch_kron64_multi(bf16_dtype, table, n_jobs, wg_per_job, o, (hipStream_t)stream), what);
}
int fq_kron_quant_multi_f16(const void* table, int n_jobs, int wg_per_job, float sig_max, float sig_min, int flags, void* stream) {
    return kron_multi_impl("fq_kron_quant_multi_f16", 0, table, n_jobs, wg_per_job, sig_max, sig_min, flags, stream);
}
int fq_kron_quant_multi_bf16(const void* table, int n_jobs, int wg_per_job, float sig_max, float sig_min, int flags, void* stream) {
    return kron_multi_impl("fq_kron_quant_multi_bf16", 1, table, n_jobs, wg_per_job, sig_max, sig_min, flags, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Prepared calls (round 5): an ARGUMENT CACHE for the three calls a deploy module makes per forward — OnlineTrans (fq_kron_quant_*,
// one clip set, packed output), Quantizer (fq_rowquant_*) and the decode-sized Linear4bit (fq_int4_skinny_linear_f16). The plan
// holds every argument that does not change between calls; fq_plan_run passes the per-call pointers (input, second input, two
// outputs, stream) — five arguments for the host binding to convert instead of eighteen, with FRESH output buffers every call.
// A plan is immutable after creation: any number of threads may run it concurrently. The plan owns nothing on the device.
struct FqPlan {
    int kind;            // 1 kron, 2 rowquant, 3 skinny linear
    int bf16;
    const void *a, *b, *c;   // kron: left, right, workspace | skinny: w_image, w_scale, bias
    int64_t rows, ws_bytes;
    int M, N, K, flags;
    float sig_max, sig_min;
};

void* fq_plan_kron(int bf16, const void* left, const void* right, int64_t rows, int M, int N, float sig_max, float sig_min, int flags,
                   void* workspace, int64_t workspace_bytes) {
    if (!left || !right || rows < 0 || (flags & (FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM)) || !(flags & FQ_OUT_PACKED)) {
        fail(FQ_EINVAL, "fq_plan_kron: left / right NULL, rows < 0, or an output set other than FQ_OUT_PACKED (flags 0x%x)", flags);
        return nullptr;
    }
    FqPlan* p = static_cast<FqPlan*>(calloc(1, sizeof(FqPlan)));
    if (!p) { fail(FQ_EINVAL, "fq_plan_kron: out of host memory"); return nullptr; }
    *p = FqPlan{1, bf16 != 0, left, right, workspace, rows, workspace_bytes, M, N, 0, flags, sig_max, sig_min};
    return p;
}
void* fq_plan_rowquant(int bf16, int64_t rows, int cols, float sig_max, float sig_min, int flags) {
    if (rows < 0 || cols < 2 || (flags & (FQ_OUT_FAKEQUANT | FQ_OUT_TRANSFORM)) || !(flags & FQ_OUT_PACKED)) {
        fail(FQ_EINVAL, "fq_plan_rowquant: rows < 0, cols < 2, or an output set other than FQ_OUT_PACKED (flags 0x%x)", flags);
        return nullptr;
    }
    FqPlan* p = static_cast<FqPlan*>(calloc(1, sizeof(FqPlan)));
    if (!p) { fail(FQ_EINVAL, "fq_plan_rowquant: out of host memory"); return nullptr; }
    *p = FqPlan{2, bf16 != 0, nullptr, nullptr, nullptr, rows, 0, 0, cols, 0, flags, sig_max, sig_min};
    return p;
}
void* fq_plan_skinny_linear(const void* w_image, const void* w_scale, const void* bias, int64_t M, int N, int K) {
    if (!w_image || !w_scale || M < 1) {
        fail(FQ_EINVAL, "fq_plan_skinny_linear: w_image / w_scale NULL or M < 1");
        return nullptr;
    }
    FqPlan* p = static_cast<FqPlan*>(calloc(1, sizeof(FqPlan)));
    if (!p) { fail(FQ_EINVAL, "fq_plan_skinny_linear: out of host memory"); return nullptr; }
    *p = FqPlan{3, 0, w_image, w_scale, bias, M, 0, 0, N, K, 0, 0.0f, 0.0f};
    return p;
}
int fq_plan_run(const void* plan, const void* in0, const void* in1, void* out0, void* out1, void* stream) {
    const FqPlan* p = static_cast<const FqPlan*>(plan);
    if (!p) return fail(FQ_EINVAL, "fq_plan_run: plan is NULL");
    void* q[1] = {out0};
    void* sc[1] = {out1};
    switch (p->kind) {
        case 1:
            return p->bf16 ? fq_kron_quant_bf16(in0, p->a, p->b, nullptr, p->rows, p->M, p->N, &p->sig_max, &p->sig_min, 1, p->flags, q, sc, nullptr,
                                                nullptr, const_cast<void*>(p->c), p->ws_bytes, stream)
                           : fq_kron_quant_f16(in0, p->a, p->b, nullptr, p->rows, p->M, p->N, &p->sig_max, &p->sig_min, 1, p->flags, q, sc, nullptr,
                                               nullptr, const_cast<void*>(p->c), p->ws_bytes, stream);
        case 2:
            return p->bf16 ? fq_rowquant_bf16(in0, p->rows, p->N, &p->sig_max, &p->sig_min, 1, p->flags, q, sc, nullptr, stream)
                           : fq_rowquant_f16(in0, p->rows, p->N, &p->sig_max, &p->sig_min, 1, p->flags, q, sc, nullptr, stream);
        case 3:
            return fq_int4_skinny_linear_f16(in0, in1, p->a, p->b, p->c, p->rows, p->N, p->K, out0, stream);
        default:
            return fail(FQ_EINVAL, "fq_plan_run: not a plan (kind %d)", p->kind);
    }
}
void fq_plan_free(void* plan) { free(plan); }

static int block_quant_impl(const char* what, int dt, const void* x, const void* P, int64_t rows, int R, int C, int transpose_out,
                            const float* sig_max, const float* sig_min, int n_clips, int flags,
                            void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                            void* stream) {
    if (!x || !P) return fail(FQ_EINVAL, "%s: x/P is NULL", what);
    FQ_NEED_ALIGN16(what, x, P);
    if (rows < 0 || R <= 0 || C <= 0) return fail(FQ_EINVAL, "%s: bad sizes", what);
    if ((R & 31) || R > 128 || (C & 1) || C > 64)
        return fail(FQ_EUNSUPPORTED, "%s: R=%d must be 32, 64, 96 or 128 and C=%d even and <= 64", what, R, C);
    FqQuantOut o;
    int rc = fill_out(what, o, sig_max, sig_min, n_clips, flags, q_out, scale_out, fq_out, y_out);
    if (rc != FQ_OK) return rc;
    if (rows == 0) return FQ_OK;
    rc = fq_launch_block(flags | dt, (const f16*)x, (const f16*)P, rows, R, C, transpose_out, o, cu_count(),
                         (hipStream_t)stream);
    return check_launch(rc, what);
}

int fq_block_quant_f16(const void* x, const void* P, int64_t rows, int R, int C, int transpose_out,
                       const float* sig_max, const float* sig_min, int n_clips, int flags,
                       void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                       void* stream) {
    return block_quant_impl("fq_block_quant_f16", 0, x, P, rows, R, C, transpose_out, sig_max, sig_min, n_clips, flags, q_out,
                            scale_out, fq_out, y_out, stream);
}

int fq_block_quant_bf16(const void* x, const void* P, int64_t rows, int R, int C, int transpose_out,
                        const float* sig_max, const float* sig_min, int n_clips, int flags,
                        void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                        void* stream) {
    return block_quant_impl("fq_block_quant_bf16", FQ_DT_BF16, x, P, rows, R, C, transpose_out, sig_max, sig_min, n_clips, flags,
                            q_out, scale_out, fq_out, y_out, stream);
}

int fq_hadamard_f16(const void* x, void* y, int64_t rows, int n, int K, const void* hadK, float scale,
                    void* stream) {
    if (!x || !y) return fail(FQ_EINVAL, "fq_hadamard_f16: x/y is NULL");
    FQ_NEED_ALIGN16("fq_hadamard_f16", x, y, hadK);
    if (rows < 0 || n <= 0 || K <= 0 || n % K) return fail(FQ_EINVAL, "fq_hadamard_f16: bad sizes n=%d K=%d", n, K);
    const int p2 = n / K;
    if (p2 & (p2 - 1)) return fail(FQ_EINVAL, "fq_hadamard_f16: n/K=%d is not a power of two", p2);
    if (K > 1 && !hadK) return fail(FQ_EINVAL, "fq_hadamard_f16: hadK is NULL with K=%d", K);
    if (rows == 0) return FQ_OK;
    int rc = fq_launch_hadamard((const f16*)x, (f16*)y, rows, n, K, (const f16*)hadK, scale, cu_count(),
                                (hipStream_t)stream);
    return check_launch(rc, "fq_hadamard_f16");
}

int fq_int4_gemm_i32(const void* x, const void* w, int64_t M, int N, int K, void* c, void* stream) {
    if (!x || !w || !c) return fail(FQ_EINVAL, "fq_int4_gemm_i32: NULL pointer");
    FQ_NEED_ALIGN16("fq_int4_gemm_i32", x, w, c);
    if (M < 0 || N <= 0 || K <= 0) return fail(FQ_EINVAL, "fq_int4_gemm_i32: bad sizes");
    if (K % 32) return fail(FQ_EINVAL, "fq_int4_gemm_i32: K=%d must be a multiple of 32", K);
    if (M == 0) return FQ_OK;
    const int rc = fq_launch_gemm_i4((const uint8_t*)x, (const uint8_t*)w, M, N, K, (int32_t*)c, nullptr, nullptr, nullptr,
                                     nullptr, (hipStream_t)stream);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "fq_int4_gemm_i32: shape M=%lld N=%d K=%d not supported", (long long)M, N, K);
    return check_launch(rc, "fq_int4_gemm_i32");
}

int fq_int4_linear_f16(const void* x, const void* x_scale, const void* w, const void* w_scale, const void* bias,
                       int64_t M, int N, int K, void* y, void* stream) {
    if (!x || !w || !y || !x_scale || !w_scale) return fail(FQ_EINVAL, "fq_int4_linear_f16: NULL pointer");
    FQ_NEED_ALIGN16("fq_int4_linear_f16", x, w, y, w_scale, bias);   // (the epilogue reads w_scale / bias with 16-byte loads)
    if (M < 0 || N <= 0 || K <= 0) return fail(FQ_EINVAL, "fq_int4_linear_f16: bad sizes");
    if (K % 32) return fail(FQ_EINVAL, "fq_int4_linear_f16: K=%d must be a multiple of 32", K);
    if (M == 0) return FQ_OK;
    const int rc = fq_launch_gemm_i4((const uint8_t*)x, (const uint8_t*)w, M, N, K, nullptr, (f16*)y, (const f16*)x_scale,
                                     (const f16*)w_scale, (const f16*)bias, (hipStream_t)stream);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "fq_int4_linear_f16: shape M=%lld N=%d K=%d not supported", (long long)M, N, K);
    return check_launch(rc, "fq_int4_linear_f16");
}

int64_t fq_int4_frag_bytes(int N, int K) { return fq_i4_frag_bytes(N, K); }

int fq_int4_to_frag(const void* w, int N, int K, void* image, void* stream) {
    if (N <= 0 || K <= 0) return fail(FQ_EINVAL, "fq_int4_to_frag: bad sizes");
    if (K % 64) return fail(FQ_EUNSUPPORTED, "fq_int4_to_frag: K=%d must be a multiple of 64", K);
    if (!w || !image) return fail(FQ_EINVAL, "fq_int4_to_frag: NULL pointer");
    return check_launch(fq_launch_i4_to_frag((const uint8_t*)w, N, K, image, cu_count(), (hipStream_t)stream), "fq_int4_to_frag");
}

int fq_int4_skinny_gemm_i32(const void* x, const void* w_image, int64_t M, int N, int K, void* c, void* stream) {
    if (M < 0 || N <= 0 || K <= 0) return fail(FQ_EINVAL, "fq_int4_skinny_gemm_i32: bad sizes");
    if (M == 0) return FQ_OK;
    if (!x || !w_image || !c) return fail(FQ_EINVAL, "fq_int4_skinny_gemm_i32: NULL pointer");
    const int rc = fq_launch_gemm_i4_skinny((const uint8_t*)x, w_image, M, N, K, (int32_t*)c, nullptr, nullptr, nullptr, nullptr,
                                            (hipStream_t)stream);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "fq_int4_skinny_gemm_i32: M=%lld K=%d (M <= 128, K %% 64 == 0)", (long long)M, K);
    return check_launch(rc, "fq_int4_skinny_gemm_i32");
}

int fq_int4_skinny_linear_f16(const void* x, const void* x_scale, const void* w_image, const void* w_scale,
                              const void* bias, int64_t M, int N, int K, void* y, void* stream) {
    if (M < 0 || N <= 0 || K <= 0) return fail(FQ_EINVAL, "fq_int4_skinny_linear_f16: bad sizes");
    if (M == 0) return FQ_OK;
    if (!x || !w_image || !y || !x_scale || !w_scale) return fail(FQ_EINVAL, "fq_int4_skinny_linear_f16: NULL pointer");
    const int rc = fq_launch_gemm_i4_skinny((const uint8_t*)x, w_image, M, N, K, nullptr, (f16*)y, (const f16*)x_scale,
                                            (const f16*)w_scale, (const f16*)bias, (hipStream_t)stream);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "fq_int4_skinny_linear_f16: M=%lld K=%d (M <= 128, K %% 64 == 0)", (long long)M, K);
    return check_launch(rc, "fq_int4_skinny_linear_f16");
}

int64_t fq_int4_skinny_split_workspace_bytes(int64_t M, int N, int K) {
    if (M < 0 || N <= 0 || K <= 0) return -1;
    return fq_gemm_i4_skinny_split_ws_bytes(M, N, K);
}

int fq_int4_skinny_linear_split_f16(const void* x, const void* x_scale, const void* w_image, const void* w_scale, const void* bias, int64_t M, int N,
                                    int K, void* y, void* workspace, int64_t workspace_bytes, void* stream) {
    const char* what = "fq_int4_skinny_linear_split_f16";
    if (M < 0 || N <= 0 || K <= 0) return fail(FQ_EINVAL, "%s: bad sizes", what);
    if (M == 0) return FQ_OK;
    if (!x || !w_image || !y || !x_scale || !w_scale) return fail(FQ_EINVAL, "%s: NULL pointer", what);
    FQ_NEED_ALIGN16(what, workspace);
    const int64_t need = fq_gemm_i4_skinny_split_ws_bytes(M, N, K);
    if (workspace && need > 0 && workspace_bytes < need)
        return fail(FQ_EINVAL, "%s: workspace of %lld bytes, fq_int4_skinny_split_workspace_bytes says %lld", what, (long long)workspace_bytes, (long long)need);
    const int rc = fq_launch_gemm_i4_skinny_split((const uint8_t*)x, w_image, M, N, K, (f16*)y, (const f16*)x_scale, (const f16*)w_scale, (const f16*)bias,
                                                  need > 0 ? (int*)workspace : nullptr, (hipStream_t)stream);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "%s: M=%lld K=%d (M <= 128, K %% 64 == 0)", what, (long long)M, K);
    return check_launch(rc, what);
}

int fq_int4_skinny_linear_multi_f16(int n, const void* const* x, const void* const* x_scale, const void* const* w_image,
                                    const void* const* w_scale, const void* const* bias, int64_t M, const int* N, int K, void* const* y,
                                    void* stream) {
    const char* what = "fq_int4_skinny_linear_multi_f16";
    if (n < 1 || n > 4) return fail(FQ_EINVAL, "%s: n=%d problems (1..4)", what, n);
    if (M < 0 || K <= 0 || !N) return fail(FQ_EINVAL, "%s: bad sizes", what);
    if (!x || !x_scale || !w_image || !w_scale || !y) return fail(FQ_EINVAL, "%s: NULL pointer table", what);
    for (int p = 0; p < n; ++p) {
        if (N[p] <= 0) return fail(FQ_EINVAL, "%s: N[%d]=%d", what, p, N[p]);
        if (M && (!x[p] || !x_scale[p] || !w_image[p] || !w_scale[p] || !y[p])) return fail(FQ_EINVAL, "%s: NULL pointer in problem %d", what, p);
    }
    if (M == 0) return FQ_OK;
    const int rc = fq_launch_gemm_i4_skinny_multi(n, (const uint8_t* const*)x, w_image, M, N, K, (f16* const*)y, (const f16* const*)x_scale,
                                                  (const f16* const*)w_scale, (const f16* const*)bias, (hipStream_t)stream);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "%s: M=%lld K=%d (M <= 128, K %% 64 == 0)", what, (long long)M, K);
    return check_launch(rc, what);
}

int fq_kron64_linear_multi_f16(const void* x, int rmsnorm, float rms_eps, const void* left, const void* right, int64_t M, int n,
                               const float* sig_max, const float* sig_min, int flags, const void* const* w_image, const void* const* w_scale,
                               const void* const* bias, const int* N, void* const* y, void* workspace, int64_t workspace_bytes, void* stream) {
    const char* what = "fq_kron64_linear_multi_f16";
    if (n < 1 || n > 4) return fail(FQ_EINVAL, "%s: n=%d problems (1..4)", what, n);
    if (M < 0 || !N) return fail(FQ_EINVAL, "%s: bad sizes", what);
    if (flags & ~(FQ_NO_CLAMP0 | FQ_ROUND_Y_F16 | FQ_WS_PREPARED)) return fail(FQ_EINVAL, "%s: unknown flag bits 0x%x", what, flags);
    if (rmsnorm && !(rms_eps >= 0.0f)) return fail(FQ_EINVAL, "%s: eps must be >= 0", what);
    if (!sig_max || !sig_min || !w_image || !w_scale || !y) return fail(FQ_EINVAL, "%s: NULL pointer table", what);
    for (int p = 0; p < n; ++p) {
        if (N[p] <= 0) return fail(FQ_EINVAL, "%s: N[%d]=%d", what, p, N[p]);
        if (!(sig_max[p] > 0.0f) || !(sig_min[p] > 0.0f)) return fail(FQ_EINVAL, "%s: sig_max/sig_min must be > 0", what);
        if (M && (!w_image[p] || !w_scale[p] || !y[p])) return fail(FQ_EINVAL, "%s: NULL pointer in problem %d", what, p);
    }
    if (M == 0) return FQ_OK;
    if (!x || !left || !right) return fail(FQ_EINVAL, "%s: x/left/right is NULL", what);
    if (!workspace || workspace_bytes < FQ_K64_WS_BYTES)
        return fail(FQ_EINVAL, "%s: workspace of %lld bytes required (fq_kron_workspace_bytes(64, 64))", what, (long long)FQ_K64_WS_BYTES);
    FQ_NEED_ALIGN16(what, x, left, right, workspace);
    if (M > 16) return fail(FQ_EUNSUPPORTED, "%s: M=%lld tokens (the fused decode launch takes 1..16; run the transform and the linears as two launches)", what, (long long)M);
    for (int p = 0; p < n; ++p)
        if (N[p] & 31) return fail(FQ_EUNSUPPORTED, "%s: N[%d]=%d is not a multiple of 32", what, p, N[p]);
    int rc;
    if (!(flags & FQ_WS_PREPARED)) {
        rc = fq_launch_kron64_prepare(left, right, workspace, (hipStream_t)stream);
        if (rc == 0) rc = fq_launch_kron_prepare((const f16*)left, (const f16*)right, 64, 64,
                                                 static_cast<unsigned char*>(workspace) + FQ_K64_IMAGE_BYTES, (hipStream_t)stream);
        if (rc != 0) return check_launch(rc, what);
    }
    rc = fq_launch_kron64_linear((const f16*)x, workspace, M, rms_eps, rmsnorm != 0, flags & (FQ_NO_CLAMP0 | FQ_ROUND_Y_F16), n, w_image,
                                 (const f16* const*)w_scale, (const f16* const*)bias, sig_max, sig_min, N, (f16* const*)y, cu_count(),
                                 (hipStream_t)stream);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "%s: shape not covered", what);
    return check_launch(rc, what);
}

int fq_int4_to_bf6(const void* q, int64_t rows, int K, int role, void* blob, void* stream) {
    if (rows < 0 || K <= 0 || (role != 0 && role != 1)) return fail(FQ_EINVAL, "fq_int4_to_bf6: bad arguments");
    if (K % 64) return fail(FQ_EUNSUPPORTED, "fq_int4_to_bf6: K=%d must be a multiple of 64", K);
    if (rows == 0) return FQ_OK;
    if (!q || !blob) return fail(FQ_EINVAL, "fq_int4_to_bf6: NULL pointer");
    const int rc = fq_launch_i4_to_bf6((const uint8_t*)q, rows, K, role, (uint8_t*)blob, cu_count(), (hipStream_t)stream);
    return check_launch(rc, "fq_int4_to_bf6");
}

int fq_bf6_gemm_i32(const void* xblob, const void* wblob, int64_t M, int N, int K, void* c, void* stream) {
    if (M < 0 || N <= 0 || K <= 0) return fail(FQ_EINVAL, "fq_bf6_gemm_i32: bad sizes");
    if (M == 0) return FQ_OK;
    if (!xblob || !wblob || !c) return fail(FQ_EINVAL, "fq_bf6_gemm_i32: NULL pointer");
    const int rc = fq_launch_gemm_bf6((const uint8_t*)xblob, (const uint8_t*)wblob, M, N, K, (int32_t*)c, nullptr, nullptr,
                                      nullptr, nullptr, (hipStream_t)stream);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "fq_bf6_gemm_i32: shape M=%lld N=%d K=%d not covered (K %% 128, N %% 16)", (long long)M, N, K);
    return check_launch(rc, "fq_bf6_gemm_i32");
}

int fq_bf6_linear_f16(const void* xblob, const void* x_scale, const void* wblob, const void* w_scale, const void* bias,
                      int64_t M, int N, int K, void* y, void* stream) {
    if (M < 0 || N <= 0 || K <= 0) return fail(FQ_EINVAL, "fq_bf6_linear_f16: bad sizes");
    if (M == 0) return FQ_OK;
    if (!xblob || !wblob || !y || !x_scale || !w_scale) return fail(FQ_EINVAL, "fq_bf6_linear_f16: NULL pointer");
    const int rc = fq_launch_gemm_bf6((const uint8_t*)xblob, (const uint8_t*)wblob, M, N, K, nullptr, (f16*)y,
                                      (const f16*)x_scale, (const f16*)w_scale, (const f16*)bias, (hipStream_t)stream);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "fq_bf6_linear_f16: shape M=%lld N=%d K=%d not covered (K %% 128, N %% 16)", (long long)M, N, K);
    return check_launch(rc, "fq_bf6_linear_f16");
}

int fq_int4_linear_fp6_f16(const void* x, const void* x_scale, const void* w, const void* wblob, const void* w_scale,
                           const void* bias, int64_t M, int N, int K, void* y, void* scratch, int64_t scratch_bytes, void* stream) {
    const char* what = "fq_int4_linear_fp6_f16";
    if (M < 0 || N <= 0 || K <= 0) return fail(FQ_EINVAL, "%s: bad sizes", what);
    if (M == 0) return FQ_OK;
    if (!x || !y || !x_scale || !w_scale || !scratch || (!w && !wblob)) return fail(FQ_EINVAL, "%s: NULL pointer", what);
    if ((K & 127) || (N & 15) || K > (1 << 18))
        return fail(FQ_EUNSUPPORTED, "%s: shape M=%lld N=%d K=%d not covered (K %% 128, N %% 16)", what, (long long)M, N, K);
    FQ_NEED_ALIGN16(what, x, w, wblob, y, scratch, w_scale, bias);
    const int64_t xb = fq_bf6_blob_bytes(M, K), wb = wblob ? 0 : fq_bf6_blob_bytes(N, K);
    if (scratch_bytes < xb + wb)
        return fail(FQ_EINVAL, "%s: scratch of %lld bytes required (got %lld)", what, (long long)(xb + wb), (long long)scratch_bytes);
    uint8_t* xs = static_cast<uint8_t*>(scratch);
    int rc = fq_launch_i4_to_bf6((const uint8_t*)x, M, K, 0, xs, cu_count(), (hipStream_t)stream);
    if (rc != 0) return check_launch(rc, what);
    const uint8_t* wsrc = (const uint8_t*)wblob;
    if (!wblob) {
        rc = fq_launch_i4_to_bf6((const uint8_t*)w, N, K, 1, xs + xb, cu_count(), (hipStream_t)stream);
        if (rc != 0) return check_launch(rc, what);
        wsrc = xs + xb;
    }
    rc = fq_launch_gemm_bf6(xs, wsrc, M, N, K, nullptr, (f16*)y, (const f16*)x_scale, (const f16*)w_scale, (const f16*)bias,
                            (hipStream_t)stream);
    return check_launch(rc, what);
}

static int linear_fp6_multi(const char* what, int n, const void* const* x, const void* const* x_scale, const void* const* w,
                            const void* const* wblob, const void* const* w_scale, const void* const* bias, int64_t M, const int* N, int K,
                            void* const* y, void* scratch, int64_t scratch_bytes, void* stream) {
    if (n < 1 || n > 4) return fail(FQ_EINVAL, "%s: 1..4 problems (got %d)", what, n);
    if (!x || !x_scale || !w || !wblob || !w_scale || !bias || !N || !y) return fail(FQ_EINVAL, "%s: NULL table", what);
    if (M < 0 || K <= 0) return fail(FQ_EINVAL, "%s: bad sizes", what);
    if (M == 0) return FQ_OK;
    if (!scratch) return fail(FQ_EINVAL, "%s: scratch is NULL", what);
    if ((K & 127) || K > (1 << 18)) return fail(FQ_EUNSUPPORTED, "%s: K=%d not covered (K %% 128)", what, K);
    int64_t need = 0;
    for (int p = 0; p < n; ++p) {
        if (N[p] <= 0 || (N[p] & 15)) return fail(FQ_EUNSUPPORTED, "%s: N[%d]=%d not covered (N %% 16)", what, p, N[p]);
        if (!x[p] || !y[p] || !x_scale[p] || !w_scale[p] || (!w[p] && !wblob[p])) return fail(FQ_EINVAL, "%s: NULL pointer in problem %d", what, p);
        FQ_NEED_ALIGN16(what, x[p], w[p], wblob[p], y[p], w_scale[p], bias[p]);
        bool seen = false;   // problems that share their activations (the same packed x) share one converted operand
        for (int q = 0; q < p; ++q) seen |= x[q] == x[p];
        if (!seen) need += fq_bf6_blob_bytes(M, K);
        if (!wblob[p]) need += fq_bf6_blob_bytes(N[p], K);
    }
    FQ_NEED_ALIGN16(what, scratch);
    if (scratch_bytes < need) return fail(FQ_EINVAL, "%s: scratch of %lld bytes required (got %lld)", what, (long long)need, (long long)scratch_bytes);
    uint8_t* cur = static_cast<uint8_t*>(scratch);
    const uint8_t* xb[4];
    const uint8_t* wb[4];
    const uint8_t* xsrc[4];   // the distinct packed activations: ONE conversion launch, their images one behind the other
    int nx = 0;
    for (int p = 0; p < n; ++p) {
        xb[p] = nullptr;
        for (int q = 0; q < p; ++q)
            if (x[q] == x[p]) xb[p] = xb[q];
        if (!xb[p]) {
            xsrc[nx++] = (const uint8_t*)x[p];
            xb[p] = cur;
            cur += fq_bf6_blob_bytes(M, K);
        }
    }
    {
        int rc = fq_launch_i4_to_bf6_multi(nx, xsrc, M, K, 0, static_cast<uint8_t*>(scratch), cu_count(), (hipStream_t)stream);
        if (rc != 0) return check_launch(rc, what);
    }
    for (int p = 0; p < n; ++p) {
        wb[p] = (const uint8_t*)wblob[p];
        if (!wb[p]) {
            int rc = fq_launch_i4_to_bf6((const uint8_t*)w[p], N[p], K, 1, cur, cu_count(), (hipStream_t)stream);
            if (rc != 0) return check_launch(rc, what);
            wb[p] = cur;
            cur += fq_bf6_blob_bytes(N[p], K);
        }
    }
    const int rc = fq_launch_gemm_bf6_multi(n, xb, wb, M, N, K, (f16* const*)y, (const f16* const*)x_scale, (const f16* const*)w_scale,
                                            (const f16* const*)bias, (hipStream_t)stream);
    return check_launch(rc, what);
}

int fq_int4_linear_fp6_multi_f16(int n, const void* const* x, const void* const* x_scale, const void* const* w, const void* const* wblob,
                                 const void* const* w_scale, const void* const* bias, int64_t M, const int* N, int K, void* const* y,
                                 void* scratch, int64_t scratch_bytes, void* stream) {
    return linear_fp6_multi("fq_int4_linear_fp6_multi_f16", n, x, x_scale, w, wblob, w_scale, bias, M, N, K, y, scratch, scratch_bytes, stream);
}

int fq_fwht_f32_f16(const void* x, void* y, int64_t vecs, int P, float scale, void* stream) {
    const char* what = "fq_fwht_f32_f16";
    if (vecs < 0 || P < 64 || P > 8192 || (P & (P - 1))) return fail(FQ_EUNSUPPORTED, "%s: P=%d (a power of two in [64, 8192])", what, P);
    if (vecs == 0) return FQ_OK;
    if (!x || !y) return fail(FQ_EINVAL, "%s: x/y is NULL", what);
    FQ_NEED_ALIGN16(what, x, y);
    return check_launch(fq_launch_fwht_f32((const f16*)x, (float*)y, vecs, P, scale, cu_count(), (hipStream_t)stream), what);
}

int fq_hadamard_quant_f16(const void* x, int64_t rows, int n, int K, const void* hadK, float scale, float sig_max,
                          float sig_min, void* q_out, void* scale_out, void* stream) {
    if (!x || !q_out || !scale_out) return fail(FQ_EINVAL, "fq_hadamard_quant_f16: NULL pointer");
    FQ_NEED_ALIGN16("fq_hadamard_quant_f16", x, hadK, q_out);
    if (rows < 0 || n <= 0 || K <= 0 || n % K) return fail(FQ_EINVAL, "fq_hadamard_quant_f16: bad sizes n=%d K=%d", n, K);
    if (K > 1 && !hadK) return fail(FQ_EINVAL, "fq_hadamard_quant_f16: hadK is NULL with K=%d", K);
    if (!(sig_max > 0.0f) || !(sig_min > 0.0f)) return fail(FQ_EINVAL, "fq_hadamard_quant_f16: sig_max/sig_min must be > 0");
    if (rows == 0) return FQ_OK;
    const int rc = fq_launch_hadamard_quant((const f16*)x, rows, n, K, (const f16*)hadK, scale, sig_max, sig_min,
                                            (uint8_t*)q_out, (f16*)scale_out, cu_count(), (hipStream_t)stream);
    if (rc == -1000)
        return fail(FQ_EUNSUPPORTED, "fq_hadamard_quant_f16: no fused kernel for n=%d K=%d (use fq_hadamard_f16 + fq_rowquant_f16)", n, K);
    return check_launch(rc, "fq_hadamard_quant_f16");
}

int fq_hadamard_quant_mfma_f16(const void* x, int64_t rows, int n, int K, const void* hadK, float scale, float sig_max,
                               float sig_min, void* q_out, void* scale_out, void* y_out, void* stream) {
    const char* what = "fq_hadamard_quant_mfma_f16";
    if (!x || !q_out != !scale_out || (!q_out && !y_out)) return fail(FQ_EINVAL, "%s: NULL pointer (x, q_out with scale_out, or y_out)", what);
    FQ_NEED_ALIGN16(what, x, hadK, q_out, y_out);
    if (rows < 0 || n <= 0 || K <= 0 || n % K) return fail(FQ_EINVAL, "%s: bad sizes n=%d K=%d", what, n, K);
    if (K > 1 && !hadK) return fail(FQ_EINVAL, "%s: hadK is NULL with K=%d", what, K);
    if (q_out && (!(sig_max > 0.0f) || !(sig_min > 0.0f))) return fail(FQ_EINVAL, "%s: sig_max/sig_min must be > 0", what);
    if (rows == 0) return FQ_OK;
    const int rc = fq_launch_had_mfma((const f16*)x, rows, n, K, (const f16*)hadK, scale, sig_max, sig_min, (uint8_t*)q_out,
                                      (f16*)scale_out, (f16*)y_out, cu_count(), (hipStream_t)stream);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "%s: n=%d K=%d (n = K * 512 or K * 1024 with K <= 32, K %% 4 == 0)", what, n, K);
    return check_launch(rc, what);
}

int fq_silu_mul_hadamard_quant_mfma_f16(const void* gate, const void* up, int64_t rows, int n, int K, const void* hadK, float scale,
                                        float sig_max, float sig_min, void* q_out, void* scale_out, void* stream) {
    const char* what = "fq_silu_mul_hadamard_quant_mfma_f16";
    if (!q_out || !scale_out) return fail(FQ_EINVAL, "%s: NULL pointer", what);
    if (rows < 0 || n <= 0 || K <= 0 || n % K) return fail(FQ_EINVAL, "%s: bad sizes n=%d K=%d", what, n, K);
    if (K > 1 && !hadK) return fail(FQ_EINVAL, "%s: hadK is NULL with K=%d", what, K);
    if (!(sig_max > 0.0f) || !(sig_min > 0.0f)) return fail(FQ_EINVAL, "%s: sig_max/sig_min must be > 0", what);
    if (rows == 0) return FQ_OK;
    if (!gate || !up) return fail(FQ_EINVAL, "%s: gate/up is NULL", what);
    FQ_NEED_ALIGN16(what, gate, up, hadK, q_out);
    const int rc = fq_launch_had_mfma((const f16*)gate, rows, n, K, (const f16*)hadK, scale, sig_max, sig_min, (uint8_t*)q_out,
                                      (f16*)scale_out, nullptr, cu_count(), (hipStream_t)stream, (const f16*)up);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "%s: n=%d K=%d (n = K * 512, K <= 32, or K * 1024, K <= 28; K %% 4 == 0)", what, n, K);
    return check_launch(rc, what);
}

int fq_hadamard_quantizer_mfma_f16(const void* x, const void* up, int64_t rows, int n, int K, const void* hadK, float scale,
                                   float input_clip_ratio, void* q_out, void* scale_out, void* y_out, void* stream) {
    const char* what = "fq_hadamard_quantizer_mfma_f16";
    if (!x || !q_out || !scale_out) return fail(FQ_EINVAL, "%s: NULL pointer (x, q_out, scale_out)", what);
    if (up && y_out) return fail(FQ_EINVAL, "%s: y_out is not available with up", what);
    if (rows < 0 || n <= 0 || K <= 0 || n % K) return fail(FQ_EINVAL, "%s: bad sizes n=%d K=%d", what, n, K);
    if (K > 1 && !hadK) return fail(FQ_EINVAL, "%s: hadK is NULL with K=%d", what, K);
    if (!(input_clip_ratio > 0.0f)) return fail(FQ_EINVAL, "%s: input_clip_ratio must be > 0", what);
    FQ_NEED_ALIGN16(what, x, up, hadK, q_out, y_out);
    if (rows == 0) return FQ_OK;
    const int rc = fq_launch_had_mfma((const f16*)x, rows, n, K, (const f16*)hadK, scale, input_clip_ratio, 1.0f, (uint8_t*)q_out,
                                      (f16*)scale_out, (f16*)y_out, cu_count(), (hipStream_t)stream, (const f16*)up, 1);
    if (rc == -1000) return fail(FQ_EUNSUPPORTED, "%s: n=%d K=%d (n = K * 512, K <= 32, or K * 1024, K <= 28; K %% 4 == 0)", what, n, K);
    return check_launch(rc, what);
}

int fq_kv_quant_f16(const void* x, const void* trans, int64_t rows, int head_dim, float clip_max, float clip_min,
                    int flags, void* q_out, void* param_out, void* y_out, void* stream) {
    if (rows < 0 || head_dim <= 0) return fail(FQ_EINVAL, "fq_kv_quant_f16: bad sizes rows=%lld head_dim=%d", (long long)rows, head_dim);
    if (flags & ~FQ_KV_LAC) return fail(FQ_EINVAL, "fq_kv_quant_f16: unknown flags 0x%x", flags);
    if (head_dim != 64 && head_dim != 128) return fail(FQ_EUNSUPPORTED, "fq_kv_quant_f16: head_dim=%d (64 or 128)", head_dim);
    if (y_out && !trans) return fail(FQ_EINVAL, "fq_kv_quant_f16: y_out without trans");
    if (rows == 0) return FQ_OK;
    if (!x || !q_out || !param_out) return fail(FQ_EINVAL, "fq_kv_quant_f16: NULL pointer");
    FQ_NEED_ALIGN16("fq_kv_quant_f16", x, trans, q_out, y_out);
    const int rc = fq_launch_kv_quant((const f16*)x, (const f16*)trans, rows, head_dim, clip_max, clip_min,
                                      (flags & FQ_KV_LAC) != 0, (uint8_t*)q_out, (f16*)param_out, (f16*)y_out, cu_count(),
                                      (hipStream_t)stream);
    return check_launch(rc, "fq_kv_quant_f16");
}

static int single_trans_impl(const char* what, int bf16_dtype, const void* x, const void* matrix, int64_t rows, int n, void* y, void* stream) {
    if (rows < 0 || n <= 0) return fail(FQ_EINVAL, "%s: bad sizes rows=%lld n=%d", what, (long long)rows, n);
    if (n != 64 && n != 128) return fail(FQ_EUNSUPPORTED, "%s: n=%d (64 or 128; smaller even n: fq_block_quant with FQ_OUT_TRANSFORM)", what, n);
    if (rows == 0) return FQ_OK;
    if (!x || !matrix || !y) return fail(FQ_EINVAL, "%s: NULL pointer", what);
    FQ_NEED_ALIGN16(what, x, matrix, y);
    return check_launch(fq_launch_rowmm(bf16_dtype, x, matrix, y, rows, n, cu_count(), (hipStream_t)stream), what);
}
int fq_single_trans_f16(const void* x, const void* matrix, int64_t rows, int n, void* y, void* stream) {
    return single_trans_impl("fq_single_trans_f16", 0, x, matrix, rows, n, y, stream);
}
int fq_single_trans_bf16(const void* x, const void* matrix, int64_t rows, int n, void* y, void* stream) {
    return single_trans_impl("fq_single_trans_bf16", 1, x, matrix, rows, n, y, stream);
}

int fq_kv_dequant_f16(const void* q, const void* param, int64_t rows, int head_dim, int flags, void* y, void* stream) {
    if (rows < 0 || head_dim <= 0 || (head_dim & 7)) return fail(FQ_EINVAL, "fq_kv_dequant_f16: bad sizes");
    if (flags & ~FQ_KV_LAC) return fail(FQ_EINVAL, "fq_kv_dequant_f16: unknown flags 0x%x", flags);
    if (rows == 0) return FQ_OK;
    if (!q || !param || !y) return fail(FQ_EINVAL, "fq_kv_dequant_f16: NULL pointer");
    FQ_NEED_ALIGN16("fq_kv_dequant_f16", q, y);
    const int rc = fq_launch_kv_dequant((const uint8_t*)q, (const f16*)param, rows, head_dim, (flags & FQ_KV_LAC) != 0,
                                        (f16*)y, cu_count(), (hipStream_t)stream);
    return check_launch(rc, "fq_kv_dequant_f16");
}

static int kv_geometry_ok(const char* what, int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, int batch) {
    if (num_layers <= 0 || layer_idx < 0 || layer_idx >= num_layers || num_heads <= 0 || page_size <= 0 || batch <= 0)
        return fail(FQ_EINVAL, "%s: bad cache geometry", what);
    if (head_dim != 64 && head_dim != 128) return fail(FQ_EUNSUPPORTED, "%s: head_dim=%d (64 or 128)", what, head_dim);
    return FQ_OK;
}

int fq_kv_append_i4(void* kv_data, void* kv_param, const void* kv_indptr, const void* kv_indices,
                    const void* last_page_offset, const void* k, const void* v, const void* k_param, const void* v_param,
                    const void* seqlen_indptr, int64_t tokens, int num_layers, int layer_idx, int num_heads, int page_size,
                    int head_dim, int batch_size, int group_size, void* stream) {
    int rc = kv_geometry_ok("fq_kv_append_i4", num_layers, layer_idx, num_heads, page_size, head_dim, batch_size);
    if (rc != FQ_OK) return rc;
    if (group_size < 1 || num_heads % group_size) return fail(FQ_EINVAL, "fq_kv_append_i4: group_size=%d must divide num_heads=%d", group_size, num_heads);
    if (tokens < 0) return fail(FQ_EINVAL, "fq_kv_append_i4: tokens < 0");
    if (!seqlen_indptr && tokens != batch_size) return fail(FQ_EINVAL, "fq_kv_append_i4: without seqlen_indptr every request appends one token (tokens == batch_size)");
    if (tokens == 0) return FQ_OK;
    if (!kv_data || !kv_param || !kv_indptr || !kv_indices || !last_page_offset || !k || !v || !k_param || !v_param)
        return fail(FQ_EINVAL, "fq_kv_append_i4: NULL pointer");
    FQ_NEED_ALIGN16("fq_kv_append_i4", kv_data, k, v);
    rc = fq_launch_kv_append(kv_data, kv_param, (const int*)kv_indptr, (const int*)kv_indices, (const int*)last_page_offset,
                             (const uint8_t*)k, (const uint8_t*)v, (const f16*)k_param, (const f16*)v_param,
                             (const int*)seqlen_indptr, tokens, num_layers, layer_idx, num_heads, page_size, head_dim, batch_size,
                             group_size, cu_count(), (hipStream_t)stream);
    return check_launch(rc, "fq_kv_append_i4");
}

int fq_kv_append_f16(void* kv_data, void* kv_param, const void* kv_indptr, const void* kv_indices,
                     const void* last_page_offset, const void* k, const void* v, const void* k_param, const void* v_param,
                     const void* seqlen_indptr, int64_t tokens, int num_layers, int layer_idx, int num_heads, int page_size,
                     int head_dim, int batch_size, int group_size, void* stream) {
    int rc = kv_geometry_ok("fq_kv_append_f16", num_layers, layer_idx, num_heads, page_size, head_dim, batch_size);
    if (rc != FQ_OK) return rc;
    if (group_size < 1 || num_heads % group_size) return fail(FQ_EINVAL, "fq_kv_append_f16: group_size=%d must divide num_heads=%d", group_size, num_heads);
    if (tokens < 0) return fail(FQ_EINVAL, "fq_kv_append_f16: tokens < 0");
    if (!seqlen_indptr && tokens != batch_size) return fail(FQ_EINVAL, "fq_kv_append_f16: without seqlen_indptr every request appends one token (tokens == batch_size)");
    if (tokens == 0) return FQ_OK;
    if (!kv_data || !kv_param || !kv_indptr || !kv_indices || !last_page_offset || !k || !v || !k_param || !v_param)
        return fail(FQ_EINVAL, "fq_kv_append_f16: NULL pointer");
    FQ_NEED_ALIGN16("fq_kv_append_f16", kv_data, k, v);
    rc = fq_launch_kv_append(kv_data, kv_param, (const int*)kv_indptr, (const int*)kv_indices, (const int*)last_page_offset,
                             (const uint8_t*)k, (const uint8_t*)v, (const f16*)k_param, (const f16*)v_param,
                             (const int*)seqlen_indptr, tokens, num_layers, layer_idx, num_heads, page_size, head_dim, batch_size,
                             group_size, cu_count(), (hipStream_t)stream, true);
    return check_launch(rc, "fq_kv_append_f16");
}

int fq_kv_batch_decode_f16_ex(void* o, const void* q, const void* q_trans, int transpose_out, const void* kv_data,
                              const void* kv_param, const void* kv_indptr, const void* kv_indices,
                              const void* last_page_offset, int num_layers, int layer_idx, int num_heads, int page_size,
                              int head_dim, int batch_size, void* stream) {
    int rc = kv_geometry_ok("fq_kv_batch_decode_f16", num_layers, layer_idx, num_heads, page_size, head_dim, batch_size);
    if (rc != FQ_OK) return rc;
    if (!o || !q || !kv_data || !kv_indptr || !kv_indices || !last_page_offset)
        return fail(FQ_EINVAL, "fq_kv_batch_decode_f16: NULL pointer");
    FQ_NEED_ALIGN16("fq_kv_batch_decode_f16", kv_data, q_trans);
    rc = fq_launch_kv_decode((f16*)o, (const f16*)q, (void*)kv_data, (void*)kv_param, (const int*)kv_indptr,
                             (const int*)kv_indices, (const int*)last_page_offset, num_layers, layer_idx, num_heads, page_size,
                             head_dim, batch_size, (const f16*)q_trans, transpose_out != 0, (hipStream_t)stream, true);
    return check_launch(rc, "fq_kv_batch_decode_f16");
}

int fq_kv_batch_decode_f16(void* o, const void* q, const void* kv_data, const void* kv_param, const void* kv_indptr,
                           const void* kv_indices, const void* last_page_offset, int num_layers, int layer_idx,
                           int num_heads, int page_size, int head_dim, int batch_size, void* stream) {
    return fq_kv_batch_decode_f16_ex(o, q, nullptr, 0, kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, num_layers,
                                     layer_idx, num_heads, page_size, head_dim, batch_size, stream);
}

int fq_kv_quant_append_i4(const void* k, const void* v, const void* trans, int64_t tokens, int src_heads, int head_dim,
                          const float* clip, int flags, void* kv_data, void* kv_param, const void* kv_indptr,
                          const void* kv_indices, const void* last_page_offset, int num_layers, int layer_idx,
                          int num_heads, int page_size, int batch_size, int group_size, void* stream) {
    int rc = kv_geometry_ok("fq_kv_quant_append_i4", num_layers, layer_idx, num_heads, page_size, head_dim, batch_size);
    if (rc != FQ_OK) return rc;
    if (flags & ~FQ_KV_LAC) return fail(FQ_EINVAL, "fq_kv_quant_append_i4: unknown flags 0x%x", flags);
    if (tokens < 0 || src_heads <= 0 || tokens % batch_size) return fail(FQ_EINVAL, "fq_kv_quant_append_i4: tokens=%lld must be a multiple of batch_size=%d", (long long)tokens, batch_size);
    if (group_size < 1 || group_size > 8 || src_heads * group_size != num_heads)
        return fail(FQ_EINVAL, "fq_kv_quant_append_i4: num_heads=%d must be src_heads=%d x group_size=%d (<= 8)", num_heads, src_heads, group_size);
    if (tokens == 0) return FQ_OK;
    if (!k || !v || !kv_data || !kv_param || !kv_indptr || !kv_indices || !last_page_offset)
        return fail(FQ_EINVAL, "fq_kv_quant_append_i4: NULL pointer");
    FQ_NEED_ALIGN16("fq_kv_quant_append_i4", k, v, trans, kv_data);
    const float unit[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    rc = fq_launch_kv_quant_append((const f16*)k, (const f16*)v, (const f16*)trans, tokens, src_heads, head_dim,
                                   clip ? clip : unit, (flags & FQ_KV_LAC) != 0, kv_data, kv_param, (const int*)kv_indptr,
                                   (const int*)kv_indices, (const int*)last_page_offset, num_layers, layer_idx, num_heads,
                                   page_size, (int)(tokens / batch_size), group_size, cu_count(), (hipStream_t)stream);
    return check_launch(rc, "fq_kv_quant_append_i4");
}

int fq_kv_batch_decode_i4_ex(void* o, const void* q, const void* q_trans, int transpose_out, const void* kv_data,
                             const void* kv_param, const void* kv_indptr, const void* kv_indices,
                             const void* last_page_offset, int num_layers, int layer_idx, int num_heads, int page_size,
                             int head_dim, int batch_size, void* stream);

int fq_kv_batch_decode_i4(void* o, const void* q, const void* kv_data, const void* kv_param, const void* kv_indptr,
                          const void* kv_indices, const void* last_page_offset, int num_layers, int layer_idx,
                          int num_heads, int page_size, int head_dim, int batch_size, void* stream) {
    return fq_kv_batch_decode_i4_ex(o, q, nullptr, 0, kv_data, kv_param, kv_indptr, kv_indices, last_page_offset, num_layers,
                                    layer_idx, num_heads, page_size, head_dim, batch_size, stream);
}

int fq_kv_batch_decode_i4_ex(void* o, const void* q, const void* q_trans, int transpose_out, const void* kv_data,
                             const void* kv_param, const void* kv_indptr, const void* kv_indices,
                             const void* last_page_offset, int num_layers, int layer_idx, int num_heads, int page_size,
                             int head_dim, int batch_size, void* stream) {
    int rc = kv_geometry_ok("fq_kv_batch_decode_i4", num_layers, layer_idx, num_heads, page_size, head_dim, batch_size);
    if (rc != FQ_OK) return rc;
    if (!o || !q || !kv_data || !kv_param || !kv_indptr || !kv_indices || !last_page_offset)
        return fail(FQ_EINVAL, "fq_kv_batch_decode_i4: NULL pointer");
    FQ_NEED_ALIGN16("fq_kv_batch_decode_i4", kv_data, q_trans);
    rc = fq_launch_kv_decode((f16*)o, (const f16*)q, (void*)kv_data, (void*)kv_param, (const int*)kv_indptr,
                             (const int*)kv_indices, (const int*)last_page_offset, num_layers, layer_idx, num_heads, page_size,
                             head_dim, batch_size, (const f16*)q_trans, transpose_out != 0, (hipStream_t)stream);
    return check_launch(rc, "fq_kv_batch_decode_i4");
}

int64_t fq_kv_decode_workspace_bytes(int batch_size, int num_heads, int head_dim) {
    if (batch_size < 0 || num_heads <= 0 || (head_dim != 64 && head_dim != 128)) return -1;
    return fq_kv_decode_ws_bytes(batch_size, num_heads, head_dim);
}

int64_t fq_kv_decode_workspace_bytes_gqa(int batch_size, int num_kv_heads, int q_group, int head_dim) {
    if (batch_size < 0 || num_kv_heads <= 0 || q_group < 1 || (head_dim != 64 && head_dim != 128)) return -1;
    return fq_kv_decode_ws_bytes_gqa(batch_size, num_kv_heads * q_group, q_group, head_dim);
}

int fq_kv_batch_decode_split(int fp16_cache, void* o, const void* q, const void* q_trans, int transpose_out, const void* kv_data,
                             const void* kv_param, const void* kv_indptr, const void* kv_indices, const void* last_page_offset,
                             int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, int batch_size, int seq_hint,
                             void* workspace, int64_t workspace_bytes, void* stream) {
    const char* what = "fq_kv_batch_decode_split";
    int rc = kv_geometry_ok(what, num_layers, layer_idx, num_heads, page_size, head_dim, batch_size);
    if (rc != FQ_OK) return rc;
    if (!o || !q || !kv_data || (!fp16_cache && !kv_param) || !kv_indptr || !kv_indices || !last_page_offset)
        return fail(FQ_EINVAL, "%s: NULL pointer", what);
    FQ_NEED_ALIGN16(what, kv_data, q_trans, workspace);
    const int64_t need = fq_kv_decode_ws_bytes(batch_size, num_heads, head_dim);
    const int splits = (workspace && need > 0) ? fq_kv_decode_splits(batch_size, num_heads, seq_hint) : 1;
    if (splits > 1 && workspace_bytes < need)
        return fail(FQ_EINVAL, "%s: workspace of %lld bytes, fq_kv_decode_workspace_bytes says %lld", what, (long long)workspace_bytes, (long long)need);
    rc = fq_launch_kv_decode((f16*)o, (const f16*)q, (void*)kv_data, (void*)kv_param, (const int*)kv_indptr, (const int*)kv_indices,
                             (const int*)last_page_offset, num_layers, layer_idx, num_heads, page_size, head_dim, batch_size,
                             (const f16*)q_trans, transpose_out != 0, (hipStream_t)stream, fp16_cache != 0, (float*)workspace, splits);
    return check_launch(rc, what);
}

int fq_kv_batch_decode_gqa(int fp16_cache, void* o, const void* q, const void* q_trans, int transpose_out, const void* kv_data,
                           const void* kv_param, const void* kv_indptr, const void* kv_indices, const void* last_page_offset,
                           int num_layers, int layer_idx, int num_kv_heads, int q_group, int page_size, int head_dim, int batch_size, int seq_hint,
                           void* workspace, int64_t workspace_bytes, void* stream) {
    const char* what = "fq_kv_batch_decode_gqa";
    int rc = kv_geometry_ok(what, num_layers, layer_idx, num_kv_heads, page_size, head_dim, batch_size);
    if (rc != FQ_OK) return rc;
    if (q_group < 1 || q_group > 64) return fail(FQ_EINVAL, "%s: q_group=%d out of [1, 64]", what, q_group);
    if (!o || !q || !kv_data || (!fp16_cache && !kv_param) || !kv_indptr || !kv_indices || !last_page_offset)
        return fail(FQ_EINVAL, "%s: NULL pointer", what);
    FQ_NEED_ALIGN16(what, kv_data, q_trans, workspace);
    const int q_heads = num_kv_heads * q_group;
    const int64_t need = fq_kv_decode_ws_bytes_gqa(batch_size, q_heads, q_group, head_dim);
    const int splits = (workspace && need > 0) ? fq_kv_decode_splits(batch_size, fq_kv_decode_wg_heads(batch_size, q_heads, q_group, head_dim), seq_hint) : 1;
    if (splits > 1 && workspace_bytes < need)
        return fail(FQ_EINVAL, "%s: workspace of %lld bytes, fq_kv_decode_workspace_bytes_gqa says %lld", what, (long long)workspace_bytes, (long long)need);
    rc = fq_launch_kv_decode((f16*)o, (const f16*)q, (void*)kv_data, (void*)kv_param, (const int*)kv_indptr, (const int*)kv_indices,
                             (const int*)last_page_offset, num_layers, layer_idx, q_heads, page_size, head_dim, batch_size,
                             (const f16*)q_trans, transpose_out != 0, (hipStream_t)stream, fp16_cache != 0, (float*)workspace, splits, q_group);
    return check_launch(rc, what);
}

int64_t fq_kv_transform_image_bytes(int head_dim) { return fq_kv_timage_bytes(head_dim); }

int fq_kv_transform_image_f16(const void* trans, int head_dim, void* image, void* stream) {
    if (head_dim != 64 && head_dim != 128) return fail(FQ_EUNSUPPORTED, "fq_kv_transform_image_f16: head_dim=%d must be 64 or 128", head_dim);
    if (!trans || !image) return fail(FQ_EINVAL, "fq_kv_transform_image_f16: NULL pointer");
    FQ_NEED_ALIGN16("fq_kv_transform_image_f16", trans, image);
    return check_launch(fq_launch_kv_timage((const f16*)trans, head_dim, image, (hipStream_t)stream), "fq_kv_transform_image_f16");
}

int fq_kv_decode_append_i4(void* o, const void* q, const void* q_trans, int transpose_out, const void* k_new, const void* v_new,
                           const void* k_trans_image, int src_heads, const void* kv_data, const void* kv_param, const void* kv_indptr,
                           const void* kv_indices, const void* last_page_offset, int num_layers, int layer_idx, int num_kv_heads, int q_group,
                           int page_size, int head_dim, int batch_size, int seq_hint, int read_one_copy, void* workspace, int64_t workspace_bytes, void* stream) {
    const char* what = "fq_kv_decode_append_i4";
    int rc = kv_geometry_ok(what, num_layers, layer_idx, num_kv_heads, page_size, head_dim, batch_size);
    if (rc != FQ_OK) return rc;
    if (q_group < 1 || q_group > 64) return fail(FQ_EINVAL, "%s: q_group=%d out of [1, 64]", what, q_group);
    if (head_dim != 128 || page_size % 16) return fail(FQ_EUNSUPPORTED, "%s: head_dim=%d, page_size=%d: needs head_dim 128 and page_size %% 16 == 0 (use fq_kv_quant_append_i4 + the decode launch)", what, head_dim, page_size);
    if (src_heads < 1 || num_kv_heads % src_heads || num_kv_heads / src_heads > 8)
        return fail(FQ_EINVAL, "%s: num_kv_heads=%d must be src_heads=%d x a group of at most 8", what, num_kv_heads, src_heads);
    if (!o || !q || !k_new || !v_new || !kv_data || !kv_param || !kv_indptr || !kv_indices || !last_page_offset)
        return fail(FQ_EINVAL, "%s: NULL pointer", what);
    FQ_NEED_ALIGN16(what, kv_data, q_trans, workspace, k_new, v_new, k_trans_image);
    const int q_heads = num_kv_heads * q_group;
    // read_one_copy: the num_kv_heads / src_heads cache heads of a source head hold identical rows (this function and fq_kv_quant_append_i4 write
    // them so): the launch reads the first copy for the whole group of query heads (and still writes the new row to every copy)
    const int copies = (read_one_copy && q_group == 1) ? num_kv_heads / src_heads : 1;
    const int share = q_group * copies;          // query heads that share one set of rows
    const int64_t need = fq_kv_decode_ws_bytes_gqa(batch_size, q_heads, share, head_dim);
    const int splits = (workspace && need > 0) ? fq_kv_decode_splits(batch_size, fq_kv_decode_wg_heads(batch_size, q_heads, share, head_dim), seq_hint) : 1;
    if (splits > 1 && workspace_bytes < need)
        return fail(FQ_EINVAL, "%s: workspace of %lld bytes, fq_kv_decode_workspace_bytes_gqa says %lld", what, (long long)workspace_bytes, (long long)need);
    rc = fq_launch_kv_decode((f16*)o, (const f16*)q, (void*)kv_data, (void*)kv_param, (const int*)kv_indptr, (const int*)kv_indices,
                             (const int*)last_page_offset, num_layers, layer_idx, q_heads, page_size, head_dim, batch_size,
                             (const f16*)q_trans, transpose_out != 0, (hipStream_t)stream, false, (float*)workspace, splits, share,
                             (const f16*)k_new, (const f16*)v_new, k_trans_image, src_heads, copies);
    return check_launch(rc, what);
}

int fq_kv_batch_decode_copies(int fp16_cache, void* o, const void* q, const void* q_trans, int transpose_out, const void* kv_data,
                              const void* kv_param, const void* kv_indptr, const void* kv_indices, const void* last_page_offset,
                              int num_layers, int layer_idx, int num_heads, int copies, int page_size, int head_dim, int batch_size, int seq_hint,
                              void* workspace, int64_t workspace_bytes, void* stream) {
    const char* what = "fq_kv_batch_decode_copies";
    int rc = kv_geometry_ok(what, num_layers, layer_idx, num_heads, page_size, head_dim, batch_size);
    if (rc != FQ_OK) return rc;
    if (copies < 1 || copies > 8 || num_heads % copies) return fail(FQ_EINVAL, "%s: copies=%d must be 1..8 and divide num_heads=%d", what, copies, num_heads);
    if (!o || !q || !kv_data || (!fp16_cache && !kv_param) || !kv_indptr || !kv_indices || !last_page_offset)
        return fail(FQ_EINVAL, "%s: NULL pointer", what);
    FQ_NEED_ALIGN16(what, kv_data, q_trans, workspace);
    const int64_t need = fq_kv_decode_ws_bytes_gqa(batch_size, num_heads, copies, head_dim);
    const int splits = (workspace && need > 0) ? fq_kv_decode_splits(batch_size, fq_kv_decode_wg_heads(batch_size, num_heads, copies, head_dim), seq_hint) : 1;
    if (splits > 1 && workspace_bytes < need)
        return fail(FQ_EINVAL, "%s: workspace of %lld bytes, fq_kv_decode_workspace_bytes_gqa(batch, num_heads / copies, copies, head_dim) says %lld", what, (long long)workspace_bytes, (long long)need);
    rc = fq_launch_kv_decode((f16*)o, (const f16*)q, (void*)kv_data, (void*)kv_param, (const int*)kv_indptr, (const int*)kv_indices,
                             (const int*)last_page_offset, num_layers, layer_idx, num_heads, page_size, head_dim, batch_size,
                             (const f16*)q_trans, transpose_out != 0, (hipStream_t)stream, fp16_cache != 0, (float*)workspace, splits, copies,
                             nullptr, nullptr, nullptr, 1, copies);
    return check_launch(rc, what);
}

static int rowquant_impl(const char* what, int dt, const void* x, int64_t rows, int cols, const float* sig_max, const float* sig_min,
                         int n_clips, int flags, void* const* q_out, void* const* scale_out,
                         void* const* fq_out, void* stream) {
    if (!x) return fail(FQ_EINVAL, "%s: x is NULL", what);
    FQ_NEED_ALIGN16(what, x);
    if (rows < 0 || cols <= 0) return fail(FQ_EINVAL, "%s: bad sizes", what);
    if (cols & 7) return fail(FQ_EUNSUPPORTED, "%s: cols=%d must be a multiple of 8", what, cols);
    if (cols > 32768) return fail(FQ_EUNSUPPORTED, "%s: cols=%d > 32768", what, cols);
    if (flags & FQ_OUT_TRANSFORM) return fail(FQ_EINVAL, "%s: FQ_OUT_TRANSFORM is meaningless here", what);
    if ((flags & FQ_ASYM) && (flags & ~(FQ_ASYM | FQ_OUT_FAKEQUANT | FQ_QUANT_F16)) )
        return fail(FQ_EINVAL, "%s: FQ_ASYM goes with FQ_OUT_FAKEQUANT (and FQ_QUANT_F16) only, flags 0x%x", what, flags);
    if ((flags & FQ_ASYM) && !(flags & FQ_OUT_FAKEQUANT)) return fail(FQ_EINVAL, "%s: FQ_ASYM needs FQ_OUT_FAKEQUANT", what);
    if ((flags & FQ_RATIO_POST) && (flags & (FQ_OUT_PACKED | FQ_OUT_FAKEQUANT | FQ_QUANT_F16 | FQ_ASYM | FQ_SIG_F16)) != (FQ_OUT_PACKED | FQ_QUANT_F16))
        return fail(FQ_EINVAL, "%s: FQ_RATIO_POST goes with FQ_OUT_PACKED | FQ_QUANT_F16 only, flags 0x%x", what, flags);
    FqQuantOut o;
    int rc = fill_out(what, o, sig_max, sig_min, n_clips, flags & ~FQ_ASYM, q_out, scale_out, fq_out, nullptr);
    if (rc != FQ_OK) return rc;
    if (rows == 0) return FQ_OK;
    rc = fq_launch_rowquant(flags | dt, (const f16*)x, rows, cols, o, cu_count(), (hipStream_t)stream);
    return check_launch(rc, what);
}

int fq_rowquant_f16(const void* x, int64_t rows, int cols, const float* sig_max, const float* sig_min,
                    int n_clips, int flags, void* const* q_out, void* const* scale_out,
                    void* const* fq_out, void* stream) {
    return rowquant_impl("fq_rowquant_f16", 0, x, rows, cols, sig_max, sig_min, n_clips, flags, q_out, scale_out, fq_out, stream);
}

int fq_rowquant_bf16(const void* x, int64_t rows, int cols, const float* sig_max, const float* sig_min,
                     int n_clips, int flags, void* const* q_out, void* const* scale_out,
                     void* const* fq_out, void* stream) {
    return rowquant_impl("fq_rowquant_bf16", FQ_DT_BF16, x, rows, cols, sig_max, sig_min, n_clips, flags, q_out, scale_out, fq_out,
                         stream);
}

static int fakequant_bits_impl(const char* what, int bf, const void* x, int64_t rows, int cols, float sig_max, float sig_min, int bits,
                               int flags, void* fq_out, void* stream) {
    if (!x || !fq_out) return fail(FQ_EINVAL, "%s: x / fq_out is NULL", what);
    FQ_NEED_ALIGN16(what, x, fq_out);
    if (rows < 0 || cols <= 0) return fail(FQ_EINVAL, "%s: bad sizes", what);
    if (cols & 7) return fail(FQ_EUNSUPPORTED, "%s: cols=%d must be a multiple of 8", what, cols);
    if (bits < 2 || bits > 8) return fail(FQ_EUNSUPPORTED, "%s: bits=%d outside [2, 8]", what, bits);
    if (flags & ~(FQ_ASYM | FQ_QUANT_F16 | FQ_SIG_F16)) return fail(FQ_EINVAL, "%s: flags 0x%x: FQ_ASYM, FQ_QUANT_F16, FQ_SIG_F16 only", what, flags);
    if ((flags & FQ_SIG_F16) && !(flags & FQ_QUANT_F16)) return fail(FQ_EINVAL, "%s: FQ_SIG_F16 needs FQ_QUANT_F16", what);
    if (!(sig_max > 0.0f) || !(sig_min > 0.0f)) return fail(FQ_EINVAL, "%s: clip factors must be positive", what);
    if (rows == 0) return FQ_OK;
    return check_launch(fq_launch_fakequant_bits(bf, x, fq_out, rows, cols, sig_max, sig_min, bits, flags, cu_count(), (hipStream_t)stream), what);
}

int fq_fakequant_bits_f16(const void* x, int64_t rows, int cols, float sig_max, float sig_min, int bits, int flags, void* fq_out,
                          void* stream) {
    return fakequant_bits_impl("fq_fakequant_bits_f16", 0, x, rows, cols, sig_max, sig_min, bits, flags, fq_out, stream);
}

int fq_fakequant_bits_bf16(const void* x, int64_t rows, int cols, float sig_max, float sig_min, int bits, int flags, void* fq_out,
                           void* stream) {
    return fakequant_bits_impl("fq_fakequant_bits_bf16", 1, x, rows, cols, sig_max, sig_min, bits, flags, fq_out, stream);
}

int fq_sym_quant_f16(const void* x, const void* scale, int64_t rows, int cols, void* q, void* stream) {
    if (!x || !scale || !q) return fail(FQ_EINVAL, "fq_sym_quant_f16: NULL pointer");
    FQ_NEED_ALIGN16("fq_sym_quant_f16", x, q);
    if (rows < 0 || cols <= 0) return fail(FQ_EINVAL, "fq_sym_quant_f16: bad sizes");
    if (rows == 0) return FQ_OK;
    return check_launch(fq_launch_sym_quant((const f16*)x, (const f16*)scale, rows, cols, (uint8_t*)q,
                                            cu_count(), (hipStream_t)stream),
                        "fq_sym_quant_f16");
}

int fq_sym_dequant_i32_f16(const void* q, const void* scale_row, const void* scale_col, int64_t rows,
                           int cols, void* x, void* stream) {
    if (!q || !scale_row || !scale_col || !x) return fail(FQ_EINVAL, "fq_sym_dequant_i32_f16: NULL pointer");
    if (rows < 0 || cols <= 0) return fail(FQ_EINVAL, "fq_sym_dequant_i32_f16: bad sizes");
    if (rows == 0) return FQ_OK;
    return check_launch(fq_launch_sym_dequant((const int32_t*)q, (const f16*)scale_row, (const f16*)scale_col,
                                              rows, cols, (f16*)x, cu_count(), (hipStream_t)stream),
                        "fq_sym_dequant_i32_f16");
}

/* debug only (not declared in fqhip.h): per-phase cycle accounting of the d=4096 packed kernel */
int fq_debug_kron64_trace(const void* x, const void* left, const void* right, int64_t rows, void* q, void* scale,
                          void* trace, void* stream) {
    FqQuantOut o;
    memset(&o, 0, sizeof(o));
    o.n_clips = 1; o.sig_max[0] = 1.0f; o.sig_min[0] = 1.0f; o.q[0] = (uint8_t*)q; o.scale[0] = (f16*)scale;
    return check_launch(fq_launch_kron64_trace((const f16*)x, (const f16*)left, (const f16*)right, rows, o,
                                               (unsigned long long*)trace, cu_count(), (hipStream_t)stream),
                        "fq_debug_kron64_trace");
}

}  // extern "C"
