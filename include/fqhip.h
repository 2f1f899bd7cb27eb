/*
 * fqhip.h — C ABI of libfqhip.so: MI355X (gfx950) implementation of FlatQuant's online
 * transform + INT4 activation-quantisation hot path.
 *
 * Every entry point:
 *   - takes raw DEVICE pointers owned by the caller (no allocation inside); every tensor (activations, factor matrices,
 *     packed / fp16 outputs, workspaces, cache pages) must start on a 16-byte boundary — the kernels use 16-byte
 *     accesses — and is refused with FQ_EINVAL otherwise (scales and index arrays: their natural alignment),
 *   - takes the HIP stream to launch on (void* == hipStream_t; NULL = default stream),
 *   - returns 0 on success or a negative FQ_E* code and never throws;
 *     fq_last_error() returns a thread-local message for the last failure,
 *   - is stateless and thread-safe (no global mutable state).
 *
 * Reference interfaces replaced (paths relative to the FlatQuant repository):
 *   fq_kron_quant_f16      deploy/kernels/kron_matmul.py:192-266 (kron_matmul) +
 *                          flatquant/flat_utils.py:6-17 (kronecker_matmul) +
 *                          flatquant/quant_utils.py:71-119 (ActivationQuantizer)
 *   fq_kron_quant_grouped_f16   flatquant/model_tools/deepseekv3_utils.py:427-452 (FlatQuantMoE.forward: rows grouped per
 *                          expert, one transform, per-expert activation quantisers) + :365-390 (_trans_forward)
 *   fq_rmsnorm_f16, fq_rmsnorm_kron_quant_f16   deploy/nn/normalization.py:4-23 (RMSNorm), alone / fused in front
 *   fq_silu_mul_f16, fq_silu_mul_kron_quant_f16, fq_silu_mul_hadamard_quant_f16, fq_silu_mul_hadamard_quant_mfma_f16
 *                          deploy/transformers/modeling_llama.py:277-279 (x_up * act_fn(x_gate) -> down_proj), alone / fused
 *   fq_block_quant_f16     deploy/kernels/block_matmul.py:231-311 (block_matmul)
 *   fq_int4_gemm_i32       deploy/kernels/gemm.cu:8-47 (matmul_host) / deploy.matmul
 *   fq_int4_linear_f16     deploy/nn/linear.py:41-56 (Linear4bit.forward = matmul + sym_dequant + bias)
 *   fq_int4_to_bf6, fq_bf6_gemm_i32, fq_bf6_linear_f16, fq_int4_linear_fp6_f16   the same two on the FP6 matrix path (bit-identical results)
 *   fq_hadamard_f16        flatquant/hadamard_utils.py:89-110,132-141 (matmul_hadU[_cuda]),
 *   (fq_hadamard_quant_f16: the same followed by deploy/nn/quantization.py:13-36, fused)
 *                          deploy/functional/online_trans.py:144-151
 *   fq_kv_quant_f16, fq_kv_dequant_f16   deploy/transformers/kv_cache.py:11-61,268 (K transform + asym INT4 pack)
 *   fq_kv_append_i4, fq_kv_batch_decode_i4   deploy/kernels/flashinfer.cu:9-96 (paged INT4 cache append + decode attention)
 *   fq_kv_append_f16, fq_kv_batch_decode_f16 deploy/kernels/flashinfer.cu:98-224 (the fp16 configuration of the same cache)
 *   fq_rowquant_f16        deploy/nn/quantization.py:13-36 (Quantizer.forward),
 *                          flatquant/quant_utils.py:77-119
 *   fq_sym_quant_f16       deploy/kernels/bindings.cpp:27-44 -> quant.cu:13-63 (sym_quant)
 *   fq_sym_dequant_i32_f16 deploy/kernels/bindings.cpp:47-87 -> quant.cu:66-101 (sym_dequant)
 */
#ifndef FQHIP_H
#define FQHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes */
#define FQ_OK            0
#define FQ_EINVAL       -1   /* bad argument (null pointer, bad size, bad flag combination) */
#define FQ_EUNSUPPORTED -2   /* shape not supported by any compiled kernel */
#define FQ_ELAUNCH      -3   /* HIP reported a launch error */

/* flags for the fused transform+quant entry points (bit-or) */
#define FQ_OUT_PACKED     0x01  /* write packed INT4 bytes + fp16 scales (deploy PackedQuantizedTensor) */
#define FQ_OUT_FAKEQUANT  0x02  /* write fp16 scale*q (FlatQuantizedLinear._eval_forward contract)      */
#define FQ_OUT_TRANSFORM  0x04  /* write the fp16 transformed activation (kronecker_matmul contract)    */
#define FQ_ROUND_Y_F16    0x08  /* round the transformed activation to fp16 BEFORE statistics/quant
                                   (path-A arithmetic: flat_utils.py:15-16 returns fp16)                */
#define FQ_NO_CLAMP0      0x10  /* do not clamp xmax>=0 / xmin<=0 (deploy Triton kernels omit the clamp,
                                   kron_matmul.py:91-100; quant_utils.py:90-91 has it)                   */
#define FQ_QUANT_F16      0x20  /* scale, x/scale and scale*q evaluated in fp16 (quant_utils.py lac=False
                                   path and quant.cu:40 __hdiv); default is fp32                         */

#define FQ_WS_PREPARED    0x40  /* `workspace` already holds the fragment image of (left, right), written by
                                   fq_kron_prepare_f16: skip the re-pack launch (the matrices are constants of a
                                   deployed layer; the re-pack is ~5 us per call)                          */

#define FQ_IN_RMSNORM     0x80  /* fq_rmsnorm_kron_quant_f16 only (set by it): x is RMS-normalised first */

#define FQ_IN_SILU_MUL    0x100 /* fq_silu_mul_* entry points only (set by them): x = fp16(up * fp16(silu(gate))) */

#define FQ_GROUP128       0x200 /* one scale per 128 CONSECUTIVE elements of the transformed token instead of one per
                                   token: ActivationQuantizer(groupsize=128) reshapes to (-1, groupsize) before it takes
                                   the extrema (vllm_custom/model_executor/layers/quantization/utils/fake_quant_utils.py:
                                   72-78); deepseek_v3/kernel.py:10-30 uses the same 128-element blocks. scale_out is then
                                   [rows, M*N/128]. Fused (one clip set, M*N % 128 == 0): packed output with fp32 arithmetic at
                                   N = 64 quantises the fp32 accumulator in the wave-per-token kernels; every other output set
                                   (fake-quant, transform, FQ_QUANT_F16), dtype (fp16 / bf16), plain or grouped launch, at 32x64,
                                   56x64, 64x64, 64x80, 64x112, 64x128 and 86..128x128 runs the GROUP EPILOGUE of the
                                   workgroup-per-token kernel (round 3) and needs FQ_ROUND_Y_F16 — it quantises the transform
                                   rounded to the activation dtype, what ActivationQuantizer(groupsize=128) is handed.
                                   FQ_EUNSUPPORTED otherwise: write FQ_OUT_TRANSFORM and run fq_rowquant_f16 with cols = 128 over
                                   the reshaped buffer (what flatquant_amd.ops does) */

#define FQ_SIG_F16        0x400 /* with FQ_QUANT_F16: extremum x sigmoid is rounded to fp16 before the division by 7 — what
                                   deploy/nn/quantization.py:21-22 evaluates (an fp16 [rows,1] tensor times a 0-dim fp32
                                   tensor gives an fp16 tensor under torch's type promotion; the (1,)-shaped parameters of
                                   flatquant/quant_utils.py:96-97 promote the product, scale and quotient to fp32 instead) */

#define FQ_ASYM           0x800 /* fq_rowquant_f16 with FQ_OUT_FAKEQUANT alone: the ASYMMETRIC quantiser of
                                   flatquant/quant_utils.py:33-46,109-117 (ActivationQuantizer(sym=False): the K / V / Q cache
                                   quantisers under --k_asym --v_asym, llama_utils.py:124-132): scale = (xmax - xmin) / 15,
                                   zero = rint(-xmin / scale), out = scale * (clamp(rint(x / scale) + zero, 0, 15) - zero);
                                   fp32 arithmetic, or fp16 throughout with FQ_QUANT_F16 (no lac / clip_ratio / half module) */

#define FQ_RATIO_POST     0x10000 /* fq_rowquant_f16 with FQ_QUANT_F16 | FQ_OUT_PACKED: the factor multiplies the SCALE, not the
                                   extrema — deploy.nn.Quantizer(input_clip_ratio) / functional.quant(input_clip_ratio),
                                   deploy/nn/quantization.py:30, deploy/functional/online_trans.py:106:
                                   scale = fp16( fp32( fp16(max|x| / 7) * ratio ) ): the product in torch's float opmath, then
                                   rounded to fp16 (what the CPU computes for an fp16 tensor times a python scalar); sig_max
                                   carries the ratio; an all-zero row gets scale 0 like the reference's (digits 0 either way).
                                   Also taken by fq_kron_quant_ex_f16 with FQ_OUT_PACKED [| FQ_OUT_TRANSFORM] | FQ_QUANT_F16 |
                                   FQ_ROUND_Y_F16 on 64 < M <= 192, N = 64 (the tall kernel: a Hadamard rotation of n = K' x 64 — 11008 —
                                   in front of Quantizer(lac=False), deploy/transformers/modeling_llama.py:244-252, as one launch);
                                   FQ_EUNSUPPORTED on every other pair; the n = K * 512 / K * 1024 widths: fq_hadamard_quantizer_mfma_f16 */

#define FQ_MAX_CLIPS 4

/*
 * bfloat16 activations: the *_bf16 entry points.
 * The reference's fake-quant path is dtype-generic (flatquant/flat_utils.py:6-17 multiplies in x's dtype;
 * flatquant/quant_utils.py:86 casts q_max to x's dtype) and its eval pipeline feeds it whatever the checkpoint declares:
 * flatquant/model_utils.py:20,34,54,68 (torch_dtype='auto': bfloat16 for Llama-3 and Qwen2.5), train_utils.py:28,
 * main_dpskv3.py:241,395 and deepseek_v3/model.py:807 (set_default_dtype(bfloat16)). fq_kron_quant_bf16,
 * fq_kron_quant_grouped_bf16, fq_block_quant_bf16, fq_rowquant_bf16 and fq_kron_prepare_bf16 take the SAME arguments as
 * their _f16 namesakes with every fp16 tensor (x, left, right, diag, P, fq_out, y_out, scale_out) a bf16 tensor, and compute
 * what torch computes for bf16 tensors: bf16 operands on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, the intermediate
 * U = x . right rounded to bf16 (torch.matmul's result type), FQ_ROUND_Y_F16 = the transformed activation rounded to bf16,
 * FQ_QUANT_F16 / FQ_SIG_F16 = scale, quotient and product (extremum x sigmoid) rounded to bf16 — the route torch's type
 * promotion takes when the clip parameters are bf16 as well (main_dpskv3.py:395) or absent — and fp32 statistics / division
 * otherwise (fp32 clip parameters next to a bf16 model: flat_linear.py:16 under the fp32 default dtype). Packed output is an
 * extension here (the reference's deploy kernels assert fp16): the scales come back in bf16. Not offered in bf16: FQ_GROUP128
 * fused, FQ_IN_RMSNORM / FQ_IN_SILU_MUL and post_scale forms (deploy-module contracts, fp16-only in the reference).
 */

/*
 * Fused Kronecker transform + per-token symmetric INT4 quantisation.
 *   y[t] = x[t] (as [M,N] row-major) ; U = fp16(x[t] . right) ; Y = left^T . U   (fp32 accumulate)
 *   == x_flat[t] @ kron(left, right)                         (flat_utils.py:6-17)
 * then per token and per clip set c:
 *   xmax = max(Y) (clamped >=0), xmin = min(Y) (clamped <=0)
 *   m = max(|xmin*sig_min[c]|, xmax*sig_max[c]) ; scale = m/7 (1 if m==0)
 *   q = clamp(rint(Y/scale), -8, 7) ; byte j = (q[2j+1] << 4) | (q[2j] & 15)
 * x      [rows, M*N] fp16 contiguous
 * left   [M, M] fp16 row-major  (hadL / matrix_left / left_matrix)
 * right  [N, N] fp16 row-major  (hadR / matrix_right / right_matrix)
 * diag   [M*N] fp16 or NULL: x is multiplied by diag (rounded to fp16) first (trans_utils.py:86-90)
 * sig_max/sig_min: HOST arrays of n_clips floats = sigmoid(clip_factor_a_max/min) (1.0f = no clipping)
 * q_out[c]     [rows, M*N/2] uint8   (FQ_OUT_PACKED)
 * scale_out[c] [rows] fp16           (FQ_OUT_PACKED)
 * fq_out[c]    [rows, M*N] fp16      (FQ_OUT_FAKEQUANT)
 * y_out        [rows, M*N] fp16      (FQ_OUT_TRANSFORM)
 * workspace   device scratch of at least fq_kron_workspace_bytes(M, N) bytes: the call re-packs left/right into MFMA fragment
 *             order there before the main kernel (skipped with FQ_WS_PREPARED). OPTIONAL for M = N = 64 (NULL / 0 bytes:
 *             the kernel gathers its fragments from the matrices itself; with the 32 KB workspace every workgroup starts
 *             with one coalesced load per thread instead of eight 2-byte gathers).
 */
int fq_kron_quant_f16(const void* x, const void* left, const void* right, const void* diag,
                      int64_t rows, int M, int N,
                      const float* sig_max, const float* sig_min, int n_clips, int flags,
                      void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                      void* workspace, int64_t workspace_bytes, void* stream);
int fq_kron_quant_bf16(const void* x, const void* left, const void* right, const void* diag,
                       int64_t rows, int M, int N,
                       const float* sig_max, const float* sig_min, int n_clips, int flags,
                       void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                       void* workspace, int64_t workspace_bytes, void* stream);

/*
 * Grouped (per-expert) form of fq_kron_quant_f16: the routed experts of a MoE layer
 * (flatquant/model_tools/deepseekv3_utils.py:427-452: `idx, top = torch.where(indices == i)`, `expert(x[idx], ...)`; the
 * rows of one expert are contiguous once the (token, expert) pairs are sorted by expert). All groups share the
 * transform (`routed_w2_trans`, deepseekv3_utils.py:470; the reference hard-codes independent_w2_trans = False) and each
 * group quantises with its own clip pair (one ActivationQuantizer per expert is what the expert module owns,
 * deepseekv3_utils.py:369-371; the reference shares one across the routed experts: pass n_groups equal pairs).
 *   x              [rows, M*N] fp16, rows sorted by group
 *   group_offsets  DEVICE int64 [n_groups + 1], non-decreasing, [0] = 0, [n_groups] = rows; empty groups allowed
 *   sig_max_g / sig_min_g   DEVICE float [n_groups] = sigmoid(clip_factor_a_max/min) of each group
 *   q_out [rows, M*N/2] u8, scale_out [rows] fp16 ([rows, M*N/128] with FQ_GROUP128), fq_out / y_out [rows, M*N] fp16,
 *   selected by flags as in fq_kron_quant_f16 (one clip set). Nothing is read back to the host: no synchronisation.
 * Shapes: every pair fq_kron_quant_f16 takes (fq_kron_workspace_bytes(M, N) >= 0); FQ_EUNSUPPORTED otherwise. workspace / FQ_WS_PREPARED as in fq_kron_quant_f16.
 */
int fq_kron_quant_grouped_f16(const void* x, const void* left, const void* right, int64_t rows, int M, int N,
                              const int64_t* group_offsets, int n_groups, const float* sig_max_g, const float* sig_min_g,
                              int flags, void* q_out, void* scale_out, void* fq_out, void* y_out,
                              void* workspace, int64_t workspace_bytes, void* stream);
int fq_kron_quant_grouped_bf16(const void* x, const void* left, const void* right, int64_t rows, int M, int N,
                               const int64_t* group_offsets, int n_groups, const float* sig_max_g, const float* sig_min_g,
                               int flags, void* q_out, void* scale_out, void* fq_out, void* y_out,
                               void* workspace, int64_t workspace_bytes, void* stream);

/*
 * The same with ONE FACTOR PAIR PER GROUP — the `routed_w2_trans[i]` branch of flatquant/model_tools/deepseekv3_utils.py:443-446
 * (independent_w2_trans: every routed expert owns its transform) — in one launch, nothing read back:
 *   left_g  [n_groups, M, M], right_g [n_groups, N, N]   contiguous device tensors of x's element type
 *   workspace  n_groups * fq_kron_workspace_bytes(M, N) bytes always suffice: one fragment image per group, written by the
 *              call unless FQ_WS_PREPARED (the matrices are constants of a deployed layer). A workgroup re-reads the image of a
 *              token's group when the group changes: once per group.
 * sig_max_g / sig_min_g may be NULL for a transform-only launch. Factor pairs: 32x64 (DeepSeek-V3 moe_inter 2048), 56x64, 64x64,
 * 64x80, 64x112 (7168), 64x128, 86..128 x 128; FQ_EUNSUPPORTED otherwise.
 */
int fq_kron_quant_grouped_mats_f16(const void* x, const void* left_g, const void* right_g, int64_t rows, int M, int N,
                                   const int64_t* group_offsets, int n_groups, const float* sig_max_g, const float* sig_min_g,
                                   int flags, void* q_out, void* scale_out, void* fq_out, void* y_out,
                                   void* workspace, int64_t workspace_bytes, void* stream);
int fq_kron_quant_grouped_mats_bf16(const void* x, const void* left_g, const void* right_g, int64_t rows, int M, int N,
                                    const int64_t* group_offsets, int n_groups, const float* sig_max_g, const float* sig_min_g,
                                    int flags, void* q_out, void* scale_out, void* fq_out, void* y_out,
                                    void* workspace, int64_t workspace_bytes, void* stream);

/*
 * deploy.nn.RMSNorm (deploy/nn/normalization.py:16-23; the weight is folded into the next layer) in front of the
 * transform, in the same launch:  x <- fp16( fp32(x) * rsqrt( sum(x^2) / (M*N) + eps ) ),  then exactly
 * fq_kron_quant_f16 on that. Saves writing and re-reading the normalised activation (4 bytes per element).
 * The fp32 sum of squares is accumulated per lane and combined across the wave (an order of its own, like any other
 * kernel's: results agree with the unfused sequence to the last fp32 bits of the variance, not bit for bit).
 * Only the 64 x 64 factor pair (d = 4096) is fused; FQ_EUNSUPPORTED otherwise: run fq_rmsnorm_f16 first then.
 */
int fq_rmsnorm_kron_quant_f16(const void* x, float eps, const void* left, const void* right, int64_t rows, int M, int N,
                              const float* sig_max, const float* sig_min, int n_clips, int flags,
                              void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                              void* stream);

/* The same for the other pairs of the supported models' hidden sizes (round 3): M <= 64 with N in {64, 80, 112, 128} — 64 x 128
 * (Llama-2-70B, 8192), 64 x 112 (DeepSeek-V3, 7168), 56 x 64 (3584), 64 x 80 (5120), 32 x 64 — packed output, in the
 * wave-per-token kernel; 64 x 64 as above. workspace / FQ_WS_PREPARED as in fq_kron_quant_f16. FQ_EUNSUPPORTED otherwise. */
int fq_rmsnorm_kron_quant_ws_f16(const void* x, float eps, const void* left, const void* right, int64_t rows, int M, int N,
                                 const float* sig_max, const float* sig_min, int n_clips, int flags,
                                 void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                                 void* workspace, int64_t workspace_bytes, void* stream);

/* deploy.nn.RMSNorm alone: y[r] = fp16( fp32(x[r]) * rsqrt( sum(x[r]^2) / cols + eps ) ), cols % 8 == 0, cols <= 16384. */
int fq_rmsnorm_f16(const void* x, void* y, int64_t rows, int cols, float eps, void* stream);

/*
 * x_up * act_fn(x_gate) (deploy/transformers/modeling_llama.py:277-278, act_fn = SiLU, fp16 tensors) in front of the
 * down_proj transform, in the same launch: the product is formed while a token is staged and never goes to HBM
 * (the eager sequence moves 10 bytes per element before the transform reads its 2).
 *   ac = fp16( g / (1 + exp(-g)) ) with g = fp32(gate);  x = fp16(ac * up);  then exactly fq_kron_quant_f16(x, ...).
 * Fused for the factor pairs with M > 64 (ffn widths: 112x128, 86..96x128, 128x224); FQ_EUNSUPPORTED otherwise:
 * run fq_silu_mul_f16 first then. Other arguments as fq_kron_quant_f16 (no diag).
 */
int fq_silu_mul_kron_quant_f16(const void* gate, const void* up, const void* left, const void* right,
                               int64_t rows, int M, int N,
                               const float* sig_max, const float* sig_min, int n_clips, int flags,
                               void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                               void* workspace, int64_t workspace_bytes, void* stream);

/*
 * fq_kron_quant_f16 with two extras, for transforms that are Kronecker products but not FlatQuant's own pair:
 *   post_scale  the transformed activation is multiplied by this fp32 factor before it is rounded / quantised: a
 *               normalisation fp16 factor matrices cannot carry exactly. The online Hadamard rotation of n = K * P
 *               (flatquant/hadamard_utils.py:132-141, deploy/functional/online_trans.py:144-151) is such a product:
 *               x.view(K*P/N, N) -> left = kron(hadK, H_{P/N}) (+-1 entries), right = H_N * 2^-e, post_scale = 2^e / sqrt(n);
 *               that is how flatquant_amd runs n = 14336 (112 x 128), 28672 (112 x 256) and 11008 (172 x 64) in front of the Quantizer.
 *   up          NULL, or the `up` tensor of the down_proj input: x is then `gate` and the transform's input is
 *               fp16(up * fp16(silu(gate))) as in fq_silu_mul_kron_quant_f16.
 * Workgroup-per-token kernels only (M > 64 or N >= 128 shapes; `up` for N >= 128 only); FQ_EUNSUPPORTED otherwise. Other arguments as
 * fq_kron_quant_f16 (no diag).
 */
int fq_kron_quant_ex_f16(const void* x, const void* up, const void* left, const void* right, int64_t rows, int M, int N,
                         float post_scale, const float* sig_max, const float* sig_min, int n_clips, int flags,
                         void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* y = fp16(up * fp16(silu(gate))) element-wise over n fp16 values (n % 8 == 0). */
int fq_silu_mul_f16(const void* gate, const void* up, void* y, int64_t n, void* stream);

/* Bytes of device workspace fq_kron_quant_f16 takes for factor sizes (M, N) (32768 for 64 x 64, where it is optional);
 * negative (FQ_EUNSUPPORTED) when no kernel handles the shape. Every pair with M <= 256, even N <= 256 and M*N <= 32768
 * has an MFMA kernel (tuned ones for the deploy shapes, csrc/fq_kron_general.hip for the rest, e.g. 128 x 148); all of them
 * but 64 x 64 read L and R from the fragment image in this workspace. */
int64_t fq_kron_workspace_bytes(int M, int N);

/* Re-pack left [M,M] / right [N,N] into the MFMA fragment image fq_kron_quant_f16 consumes, once, for callers whose
 * matrices do not change between calls (deploy/nn/online_trans.py:18-67 keeps them as buffers). Pass the same
 * workspace with FQ_WS_PREPARED afterwards. M = N = 64: writes the optional 32 KB of images when a workspace is given, a no-op
 * (FQ_OK) with workspace == NULL. */
int fq_kron_prepare_f16(const void* left, const void* right, int M, int N, void* workspace, int64_t workspace_bytes,
                        void* stream);
int fq_kron_prepare_bf16(const void* left, const void* right, int M, int N, void* workspace, int64_t workspace_bytes,
                         void* stream);

/*
 * Prepared calls (round 5): an argument cache for the calls a deploy module makes on every forward — what the reference's modules
 * do per call in Python around their kernels (deploy/nn/online_trans.py:61-99 -> functional/online_trans.py:113-122 ->
 * kernels/kron_matmul.py:192-266; deploy/nn/quantization.py:13-36; deploy/nn/linear.py:41-56). A plan fixes everything that does not
 * change between calls; fq_plan_run(plan, in0, in1, out0, out1, stream) supplies the per-call pointers:
 *   fq_plan_kron            in0 = x [rows, M*N]            out0 = q [rows, M*N/2] uint8, out1 = scale [rows]   (FQ_OUT_PACKED, one clip set;
 *                           left / right / workspace as for fq_kron_quant_{f16,bf16}; with FQ_WS_PREPARED the image fq_kron_prepare_* wrote)
 *   fq_plan_rowquant        in0 = x [rows, cols]           out0 = q, out1 = scale                              (FQ_OUT_PACKED, one clip set)
 *   fq_plan_skinny_linear   in0 = packed x [M, K/2], in1 = x scales [M] fp16      out0 = y [M, N] fp16         (fq_int4_skinny_linear_f16)
 * sig_max / sig_min are the sigmoid-ed factors (as the float arrays of the plain entry points). Outputs are the CALLER'S buffers, fresh
 * or reused; results are bit for bit those of the plain entry point. A plan is immutable and holds no device memory: run it from any
 * thread; it stays valid while the pointers it was built from do. fq_plan_* return NULL on bad arguments (fq_last_error()).
 */
void* fq_plan_kron(int bf16, const void* left, const void* right, int64_t rows, int M, int N, float sig_max, float sig_min, int flags,
                   void* workspace, int64_t workspace_bytes);
void* fq_plan_rowquant(int bf16, int64_t rows, int cols, float sig_max, float sig_min, int flags);
void* fq_plan_skinny_linear(const void* w_image, const void* w_scale, const void* bias, int64_t M, int N, int K);
int fq_plan_run(const void* plan, const void* in0, const void* in1, void* out0, void* out1, void* stream);
void fq_plan_free(void* plan);

/*
 * Multi-job launch (round 4): SEVERAL independent 64 x 64 transform + quantisation jobs — one per layer of a model, or one per
 * shard — as ONE kernel launch. What a caller that shards the rows over GPUs is left with per layer is a short launch (2048
 * tokens = 5 us of kernel behind a launch that costs as much); the jobs of a step are independent (flat_linear.py:75-80 and
 * deploy/nn/online_trans.py:61-99 hold no state between layers' inputs), so their launches fold into one. Every job has its own
 * activations, factor pair (as the prepared image fq_kron_prepare_f16 writes: `workspace`, >= fq_kron_workspace_bytes(64, 64)),
 * outputs and row count; all jobs share the clip pair and the flags (FQ_OUT_PACKED, optionally FQ_NO_CLAMP0). Results are bit
 * for bit those of n_jobs calls of fq_kron_quant_f16(..., FQ_OUT_PACKED | FQ_WS_PREPARED | flags).
 *   fq_kron_multi_table_bytes(n_jobs)   bytes of the DEVICE job table (16-byte aligned)
 *   fq_kron_multi_prepare(jobs, n_jobs, table, table_bytes, stream)   writes the table from the HOST array `jobs` (a copy on
 *       `stream`, waited for: a set-up call; `jobs` may be released on return); returns the number of workgroups per job (> 0)
 *       or a negative error code
 *   fq_kron_quant_multi_{f16,bf16}(table, n_jobs, wg_per_job, sig_max, sig_min, flags, stream)   the launch; wg_per_job must be
 *       the value fq_kron_multi_prepare returned for this table (its per-job token split is baked into the table): FQ_EINVAL otherwise
 * The table is valid for as long as the jobs' pointers and row counts are (a deployed model's layers: prepare once).
 */
typedef struct FqKronJob {
    const void* x;          /* [rows, 4096] activations */
    const void* workspace;  /* prepared image of this job's (left, right): fq_kron_prepare_f16 */
    void* q;                /* [rows, 2048] packed INT4 */
    void* scale;            /* [rows] */
    int64_t rows;
} FqKronJob;
int64_t fq_kron_multi_table_bytes(int n_jobs);
int fq_kron_multi_prepare(const FqKronJob* jobs, int n_jobs, void* table, int64_t table_bytes, void* stream);
int fq_kron_quant_multi_f16(const void* table, int n_jobs, int wg_per_job, float sig_max, float sig_min, int flags, void* stream);
int fq_kron_quant_multi_bf16(const void* table, int n_jobs, int wg_per_job, float sig_max, float sig_min, int flags, void* stream);

/*
 * Single-matrix transform over the LAST axis of [rows, R, C] blocks (o_proj head transform):
 *   Y[t] = x[t] ([R,C] row-major) . P ([C,C]);  quantised per token over all R*C values.
 * Packed output follows block_matmul.py:86-101: the quantised block is TRANSPOSED before packing when
 * transpose_out != 0 (logical [C, R] per token), natural [R, C] otherwise.
 * R (head_dim) in {32, 64, 96, 128}; C (num_attention_heads) any even number <= 64: 32 / 64 (Llama) on the tuned kernel,
 * 28 / 40 / 48 / 12 / 14 / 16 (Qwen2.5, Llama-2-13B) on the masked one — the reference kernel masks arbitrary sizes
 * (block_matmul.py:56-66,101-103). FQ_EUNSUPPORTED otherwise.
 */
int fq_block_quant_f16(const void* x, const void* P, int64_t rows, int R, int C, int transpose_out,
                       const float* sig_max, const float* sig_min, int n_clips, int flags,
                       void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                       void* stream);
int fq_block_quant_bf16(const void* x, const void* P, int64_t rows, int R, int C, int transpose_out,
                        const float* sig_max, const float* sig_min, int n_clips, int flags,
                        void* const* q_out, void* const* scale_out, void* const* fq_out, void* y_out,
                        void* stream);

/*
 * INT4 x INT4 -> INT32 GEMM on packed nibbles (deploy/kernels/gemm.cu:8-47 behind deploy.matmul,
 * deploy/__init__.py:37-41):  c[m][n] = sum_k x[m][k] * w[n][k].
 *   x [M, K/2] uint8 (packed activations, e.g. PackedQuantizedTensor.quantized_x), w [N, K/2] uint8 (Linear4bit.weight),
 *   low nibble = even k, two's complement;  c [M, N] int32.  K % 32 == 0 (the reference's assert).
 */
int fq_int4_gemm_i32(const void* x, const void* w, int64_t M, int N, int K, void* c, void* stream);

/*
 * Linear4bit.forward in one launch (deploy/nn/linear.py:41-56): the GEMM above with sym_dequant (quant.cu:66-85:
 * y = x_scale[m] * w_scale[n] * half(int(c / 10.0f)) * half(10), fp16 products left to right) and the optional bias
 * fused into its epilogue: the int32 matrix never goes to HBM.  Bit-identical to fq_int4_gemm_i32 followed by
 * fq_sym_dequant_i32_f16 (+ bias).
 *   x_scale [M] fp16, w_scale [N] fp16, bias [N] fp16 or NULL, y [M, N] fp16
 */
int fq_int4_linear_f16(const void* x, const void* x_scale, const void* w, const void* w_scale, const void* bias,
                       int64_t M, int N, int K, void* y, void* stream);

/*
 * Decode-sized batches (M <= 128): the same GEMM / Linear4bit as a weight-streaming kernel. The weights are read from an
 * image in MFMA fragment order, built once per layer:
 *   fq_int4_frag_bytes(N, K)                 bytes of the image (K % 64 == 0; -1 otherwise)
 *   fq_int4_to_frag(w [N, K/2], N, K, image)
 *   fq_int4_skinny_gemm_i32 / fq_int4_skinny_linear_f16: as fq_int4_gemm_i32 / fq_int4_linear_f16 with the image in
 *   place of w; M <= 128, K % 64 == 0, FQ_EUNSUPPORTED otherwise. Bit-identical results.
 */
int64_t fq_int4_frag_bytes(int N, int K);
int fq_int4_to_frag(const void* w, int N, int K, void* image, void* stream);
int fq_int4_skinny_gemm_i32(const void* x, const void* w_image, int64_t M, int N, int K, void* c, void* stream);
int fq_int4_skinny_linear_f16(const void* x, const void* x_scale, const void* w_image, const void* w_scale,
                              const void* bias, int64_t M, int N, int K, void* y, void* stream);

/* (round 6, third session) fq_int4_skinny_linear_f16 with the K range of every 32-feature tile split over 2 (4) workgroups where the unsplit launch
 * leaves CUs idle (at most 128 feature tiles) and either is busy with more than one token tile (33 .. 128 rows) or has a long K range (K >= 8192 from
 * 9 rows, K >= 16384 from one): the workgroups of a tile leave int32 partial
 * sums in `workspace`, the last to arrive adds them and runs sym_dequant + bias. Integer sums: y is bit-identical to the unsplit launch.
 * workspace: fq_int4_skinny_split_workspace_bytes(M, N, K) bytes (0: this geometry is not split — the plain launch runs), 16-byte aligned, ZEROED
 * once by the caller before its first use (the launch leaves its counters at zero), used by one launch at a time in stream order. NULL: never split. */
int64_t fq_int4_skinny_split_workspace_bytes(int64_t M, int N, int K);
int fq_int4_skinny_linear_split_f16(const void* x, const void* x_scale, const void* w_image, const void* w_scale, const void* bias, int64_t M, int N,
                                    int K, void* y, void* workspace, int64_t workspace_bytes, void* stream);
/*
 * (round 5) Up to FOUR decode-sized Linear4bit problems that share M (<= 128) and K — q / k / v, or up / gate, of one decoder layer
 * (deploy/nn/linear.py:40-54 runs once per projection; modeling_llama.py:66-78, 268-280), each with its own packed activations,
 * weight image (fq_int4_to_frag), scales, bias and output — as ONE launch of the weight-streaming kernel: the feature tiles of the
 * problems side by side in the grid. Bit-identical to n calls of fq_int4_skinny_linear_f16 (a decode-sized launch costs ~4 us whatever
 * it streams: three launches are three of those). Tables of n pointers; bias may be NULL (no problem has one) or hold NULL entries.
 */
int fq_int4_skinny_linear_multi_f16(int n, const void* const* x, const void* const* x_scale, const void* const* w_image,
                                    const void* const* w_scale, const void* const* bias, int64_t M, const int* N, int K, void* const* y,
                                    void* stream);

/*
 * (round 6) THE TRANSFORM AS THE GEMM'S PROLOGUE, decode regime. ONE launch for what the reference runs as ln_trans -> quantizer -> q / k / v
 * (or up / gate) projections on 1..16 tokens of d = 4096 (deploy/transformers/modeling_llama.py:66-78,268-280; deploy/nn/online_trans.py,
 * quantization.py, linear.py:40-54): [RMSNorm, normalization.py:16-23, when rmsnorm != 0] + the 64 x 64 Kronecker transform + per-token INT4
 * quantisation with problem p's clip pair (sig_max[p], sig_min[p]) + Linear4bit of problem p — n in 1..4 problems that share x and the
 * factor pair and have their own clip pair, weight image (fq_int4_to_frag), weight scales, bias (table or entries may be NULL) and output
 * y[p] [M, N[p]] fp16. The quantised activations are never written to memory: every workgroup repeats the transform of the M tokens while
 * its weights are in flight, so the launch costs ONE small-dispatch latency floor instead of two (profiles/r06_fused_decode.txt).
 * Bit-identical to fq_[rmsnorm_]kron_quant_f16(FQ_OUT_PACKED | flags, n clip sets) followed by fq_int4_skinny_linear_multi_f16.
 *   flags: FQ_NO_CLAMP0, FQ_ROUND_Y_F16, FQ_WS_PREPARED (workspace holds the images of fq_kron_prepare_f16(left, right, 64, 64)); without
 *   FQ_WS_PREPARED they are written first. workspace_bytes >= fq_kron_workspace_bytes(64, 64), required.
 *   FQ_EUNSUPPORTED: M > 16 or N[p] % 32 != 0 — run the two launches.
 */
int fq_kron64_linear_multi_f16(const void* x, int rmsnorm, float rms_eps, const void* left, const void* right, int64_t M, int n,
                               const float* sig_max, const float* sig_min, int flags, const void* const* w_image, const void* const* w_scale,
                               const void* const* bias, const int* N, void* const* y, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * The same GEMM / Linear4bit on the FP6 matrix path (v_mfma_scale_f32_32x32x64_f8f6f4, both operands BF6 = E3M2, unit
 * block scales): every integer in [-8, 7] is a BF6 value and the fp32 accumulator holds the exact integer sum
 * (K <= 2^18), so the results are bit-identical to fq_int4_gemm_i32 / fq_int4_linear_f16 — at ~1.4x the sustained
 * matrix rate of the int8 instruction and without unpack arithmetic. Operands are converted first:
 *   fq_bf6_blob_bytes(rows, K)            bytes of the converted operand (K % 64 == 0; -1 otherwise)
 *   fq_int4_to_bf6(q [rows, K/2], rows, K, role, blob)   role 0 = activations (x), 1 = weights (w; once per layer)
 * The blob layout is private to this library (fragment order of the MFMA operand, csrc/fq_gemm_bf6.hip).
 *   fq_bf6_gemm_i32 / fq_bf6_linear_f16: as the int4 entry points with blobs in place of x and w;
 *   K % 128 == 0, N % 16 == 0, K <= 262144, FQ_EUNSUPPORTED otherwise (use the int4 entry points).
 */
int64_t fq_bf6_blob_bytes(int64_t rows, int K);
int fq_int4_to_bf6(const void* q, int64_t rows, int K, int role, void* blob, void* stream);
int fq_bf6_gemm_i32(const void* xblob, const void* wblob, int64_t M, int N, int K, void* c, void* stream);
int fq_bf6_linear_f16(const void* xblob, const void* x_scale, const void* wblob, const void* w_scale, const void* bias,
                      int64_t M, int N, int K, void* y, void* stream);
/*
 * Linear4bit.forward (deploy/nn/linear.py:41-56) on the FP6 path as ONE call — what the host side needs per layer call: the packed
 * activations x [M, K/2] are converted into `scratch`, the weights are taken from the kept image `wblob` or, when it is NULL,
 * converted from the packed `w` [N, K/2] into `scratch` as well (a transient image: nothing stays resident), then
 * fq_bf6_linear_f16 runs on the two blobs — three launches on `stream`, one library call.
 *   scratch_bytes >= fq_bf6_blob_bytes(M, K) + (wblob ? 0 : fq_bf6_blob_bytes(N, K)), 16-byte aligned; contents undefined afterwards.
 * Same shape limits and the same bits as fq_bf6_linear_f16 / fq_int4_linear_f16.
 */
int fq_int4_linear_fp6_f16(const void* x, const void* x_scale, const void* w, const void* wblob, const void* w_scale,
                           const void* bias, int64_t M, int N, int K, void* y, void* scratch, int64_t scratch_bytes, void* stream);

/*
 * (round 4) Up to FOUR Linear4bit problems that share their token count M and their K — q / k / v, or up / gate, of one decoder layer:
 * each with its own quantised activations, weights, scales, bias and output (deploy/nn/linear.py:40-54 runs once per projection) — as
 * ONE GEMM launch: the feature tiles of the problems are laid side by side in the persistent workgroups' tile sequence. At 2048 tokens a
 * 4096-wide projection is 128 tiles of 256 x 256 on 256 CUs; three launches leave half the chip idle three times. Bit-identical to n
 * calls of fq_int4_linear_fp6_f16.
 *   n in 1..4; tables of n pointers: x[p] packed [M, K/2], x_scale[p] [M] fp16, w[p] packed [N[p], K/2] or NULL with wblob[p] its FP6
 *   image (one of the two), w_scale[p] [N[p]] fp16, bias[p] [N[p]] fp16 or NULL, y[p] [M, N[p]] fp16. Problems whose x[p] pointers are
 *   equal share one converted operand.
 *   scratch_bytes >= sum over DISTINCT x of fq_bf6_blob_bytes(M, K) + sum over problems without a wblob of fq_bf6_blob_bytes(N[p], K).
 *   K % 128 == 0, N[p] % 16 == 0 (FQ_EUNSUPPORTED otherwise).
 */
int fq_int4_linear_fp6_multi_f16(int n, const void* const* x, const void* const* x_scale, const void* const* w, const void* const* wblob,
                                 const void* const* w_scale, const void* const* bias, int64_t M, const int* N, int K, void* const* y,
                                 void* scratch, int64_t scratch_bytes, void* stream);

/*
 * Normalised Hadamard transform over the last axis, n = K * 2^p:
 *   y = hadK [K,K] @ FWHT_{n/K}( x.view(rows, K, n/K) ) * scale         (hadamard_utils.py:132-141)
 * fp32 butterflies, result of the FWHT rounded to fp16 before the K-factor (as the un-vendored
 * fast_hadamard_transform + fp16 matmul sequence does), output fp16.
 * hadK may be NULL iff K == 1.  In-place (y == x) is allowed.
 */
int fq_hadamard_f16(const void* x, void* y, int64_t rows, int n, int K, const void* hadK, float scale,
                    void* stream);

/*
 * OnlineTrans(force_fp32=True) (deploy/nn/online_trans.py:55-59): the reference up-casts the activation and runs
 * fast_hadamard_transform in fp32 — y [vecs, P] fp32 = FWHT_P( x [vecs, P] fp16 ) * scale, every butterfly and the scaling in fp32, NO
 * rounding to fp16 (the stage order of hadamard_utils.py:94-101; bit for bit the oracle's fwht_f32 * scale). P a power of two in
 * [64, 8192]. For n = K * P the caller views x as [rows * K, P] and applies the fp32 K x K factor itself, as the reference does
 * (`hadK.to(input.dtype) @ input`, a plain fp32 GEMM: deploy/functional/online_trans.py:148-150).
 */
int fq_fwht_f32_f16(const void* x, void* y, int64_t vecs, int P, float scale, void* stream);

/*
 * The same Hadamard transform fused with deploy.nn.Quantizer (deploy/nn/quantization.py:13-36 with lac clip
 * factors): the fp16 result never goes to HBM. Per row: extrema of the fp16 transform output (clamped through 0),
 * scale = fp16(max(|xmin*sig_min|, xmax*sig_max) / 7), q = clamp(rint(y /h scale), -8, 7) with the fp16 division
 * of quant.cu:40, packed two per byte (low nibble = even column).  Equals fq_hadamard_f16 followed by
 * fq_rowquant_f16(FQ_OUT_PACKED | FQ_QUANT_F16) bit for bit.  Returns FQ_EUNSUPPORTED for shapes the fused
 * kernels do not cover (n/K not in {512, 1024} for K > 1, n not in 512..8192 for K == 1): call the two
 * separately then.
 *   q_out [rows, n/2] uint8, scale_out [rows] fp16
 */
int fq_hadamard_quant_f16(const void* x, int64_t rows, int n, int K, const void* hadK, float scale,
                          float sig_max, float sig_min, void* q_out, void* scale_out, void* stream);

/*
 * The same rotation (+ Quantizer) with the STRUCTURE of the rotation on the matrix pipe (round 4, fq_had_mfma.hip):
 * H_512 = H_4 (x) H_4 (x) H_32 — two in-register butterflies and two K = 32 contractions (H_32, hadK) per tile, 16 MFMAs per wave and
 * token where the dense Kronecker launch of the same rotation (fq_kron_quant_ex_f16 on (hadK (x) H_4) (x) H_128) needs 60.
 * Same contracts as fq_hadamard_f16 (y_out) and fq_hadamard_quant_f16 (q_out + scale_out) EXCEPT bit identity: the intermediate is
 * rounded to fp16 at different points, the rotated values agree with fq_hadamard_f16 within the op's tolerance class (1e-3 of the
 * row maximum against the exact rotation; hadamard_utils.py:89-110 is the oracle), so a scale can differ by an fp16 step and a digit
 * by +-1 on ~1e-3 of the elements.
 * Covers n = K * 512 with 4 <= K <= 32 and n = K * 1024 with 4 <= K <= 28, K % 4 == 0 (14336 = 28 * 512: Llama-3-8B ffn; 28672 = 28 * 1024: Llama-2-70B
 * ffn, H_1024 = H_8 (x) H_4 (x) H_32, 32 MFMAs per wave and token where the dense 112 x 256 pair needs 128); FQ_EUNSUPPORTED otherwise.
 *   hadK [K, K] fp16 (+-1); q_out [rows, n/2] uint8 with scale_out [rows] fp16, or both NULL; y_out [rows, n] fp16 or NULL (y_out == x
 *   is allowed); at least one output. No workspace.
 */
int fq_hadamard_quant_mfma_f16(const void* x, int64_t rows, int n, int K, const void* hadK, float scale,
                               float sig_max, float sig_min, void* q_out, void* scale_out, void* y_out, void* stream);

/*
 * fq_hadamard_quant_mfma_f16 (packed output) on x = fp16(up * fp16(silu(gate))): deploy/transformers/modeling_llama.py:277-279 in front
 * of down_proj = Sequential(OnlineTrans(had), Quantizer, Linear4bit) (:248-253) as ONE launch on the structured kernel. Every wave
 * loads the chunks of `up` that match its own LDS-DMA instructions of `gate` and rewrites its slots of the staged token in place
 * (the arithmetic of fq_silu_mul_f16, bit for bit: the result equals fq_hadamard_quant_mfma_f16 on fq_silu_mul_f16's output).
 * Same shapes, same FQ_EUNSUPPORTED otherwise. gate, up [rows, n] fp16.
 */
int fq_silu_mul_hadamard_quant_mfma_f16(const void* gate, const void* up, int64_t rows, int n, int K, const void* hadK, float scale,
                                        float sig_max, float sig_min, void* q_out, void* scale_out, void* stream);

/*
 * The structured rotation in front of deploy.nn.Quantizer(input_clip_ratio, lac=False) — what the reference's deploy model builds under
 * options.trans == "had": down_proj = Sequential(OnlineTrans(had), Quantizer(lac=False), Linear4bit), deploy/transformers/modeling_llama.py:241-253
 * — as ONE launch: scale = fp16(max|y| / 7) * input_clip_ratio (deploy/nn/quantization.py:30; FQ_RATIO_POST arithmetic, as fq_rowquant_f16), NO
 * zero guard (an all-zero token stores scale 0 and digits 0), digits by quant.cu:40 as every Quantizer route.
 * up != NULL: x is x_gate, the rotation's input fp16(up * fp16(silu(x))) (as fq_silu_mul_hadamard_quant_mfma_f16; y_out must be NULL).
 * y_out (optional, up == NULL): the rotated activation the digits were taken from. Shapes as fq_hadamard_quant_mfma_f16.
 */
int fq_hadamard_quantizer_mfma_f16(const void* x, const void* up, int64_t rows, int n, int K, const void* hadK, float scale,
                                   float input_clip_ratio, void* q_out, void* scale_out, void* y_out, void* stream);

/* fq_hadamard_quant_f16 on x = fp16(up * fp16(silu(gate))) formed in registers (see fq_silu_mul_kron_quant_f16). */
int fq_silu_mul_hadamard_quant_f16(const void* gate, const void* up, int64_t rows, int n, int K, const void* hadK,
                                   float scale, float sig_max, float sig_min, void* q_out, void* scale_out,
                                   void* stream);

/*
 * Per-token scale + INT4 quantisation of an fp16 matrix (Quantizer.forward / ActivationQuantizer).
 * Same statistics/flags as fq_kron_quant_f16 with Y = x.  cols even, cols <= 65536.
 */
int fq_rowquant_f16(const void* x, int64_t rows, int cols,
                    const float* sig_max, const float* sig_min, int n_clips, int flags,
                    void* const* q_out, void* const* scale_out, void* const* fq_out,
                    void* stream);
int fq_rowquant_bf16(const void* x, int64_t rows, int cols,
                     const float* sig_max, const float* sig_min, int n_clips, int flags,
                     void* const* q_out, void* const* scale_out, void* const* fq_out,
                     void* stream);

/*
 * ActivationQuantizer.fake_quant with bits != 4 (round 4): flatquant/quant_utils.py:10-16 (get_qmin_qmax makes the grid a parameter —
 * --a_bits / --q_bits / --k_bits / --v_bits, args_utils.py:38,101,108,116), :76-119 (statistics), :18-46 (quantise / dequantise).
 * Fake-quant output only (the packed format of this library is the INT4 one); one clip pair; 2 <= bits <= 8.
 *   symmetric (default): qmax = 2^(bits-1) - 1, scale = max(|xmin sig_min|, xmax sig_max) / qmax (1 if 0), q = clamp(rint(x / scale), -qmax-1, qmax)
 *   FQ_ASYM:             qmax = 2^bits - 1, (0, 0) -> (-1, +1), scale = (xmax - xmin) / qmax, zero = rint(-xmin / scale),
 *                        q = clamp(rint(x / scale) + zero, 0, qmax), out = scale (q - zero)
 *   extrema through 0 in both (quant_utils.py:90-91). Default arithmetic: fp32 (lac with fp32 clip parameters). FQ_QUANT_F16: every
 *   operation rounds to the activation dtype (no lac / clip_ratio / a module cast to it); FQ_SIG_F16 with it: the extremum x factor
 *   product too (always with FQ_ASYM). bits == 4 gives the same values as fq_rowquant_* with FQ_OUT_FAKEQUANT (which is the fast path).
 *   x, fq_out [rows, cols] fp16 (bf16), cols % 8 == 0, 16-byte aligned; fq_out == x is allowed.
 */
int fq_fakequant_bits_f16(const void* x, int64_t rows, int cols, float sig_max, float sig_min, int bits, int flags,
                          void* fq_out, void* stream);
int fq_fakequant_bits_bf16(const void* x, int64_t rows, int cols, float sig_max, float sig_min, int bits, int flags,
                           void* fq_out, void* stream);

/*
 * KV-cache quantisation (deploy/transformers/kv_cache.py:11-51 asym_quantize_and_pack_i4, :268 the K transform), one
 * launch: per row of head_dim fp16 values (a row = one head of one token; x is [rows, head_dim] contiguous)
 *   y = trans ? fp16(x . trans) : x                               trans [head_dim, head_dim] fp16 row-major or NULL
 *   default (the cache's own calls, lac=False):  scale = fp16(max(fp16(ymax - ymin), 1e-5) / 15), zero = -ymin,
 *                                                q = clamp(rint(fp16(fp16(y + zero) / scale)), 0, 15)
 *   FQ_KV_LAC: extrema through 0, times clip_max / clip_min (already sigmoid-ed, fp16 values), (0,0) -> (-1,1),
 *              scale = fp16(fp16(ymax - ymin) / 15), zero = rint(fp16(-ymin / scale)),
 *              q = clamp(fp16(rint(fp16(y / scale)) + zero), 0, 15)
 * every step in fp16 arithmetic as the torch expression evaluates it.
 *   q_out [rows, head_dim/2] uint8 (low nibble = even column), param_out [rows, 2] fp16 (scale, zero) — the k_param /
 *   v_param layout of kv_cache.py:296-297; y_out [rows, head_dim] fp16 or NULL (the transformed rows, trans only).
 * head_dim in {64, 128}.
 */
#define FQ_KV_LAC 0x1
int fq_kv_quant_f16(const void* x, const void* trans, int64_t rows, int head_dim, float clip_max, float clip_min,
                    int flags, void* q_out, void* param_out, void* y_out, void* stream);

/*
 * {SVD,Inv}SingleTransMatrix.forward at n = head_dim (flatquant/trans_utils.py:21-25, 136-151; the fake-quant eval path applies
 * kcache_trans / vcache_trans to [.., heads, head_dim] activations on every forward, llama_utils.py:181-199):
 *   y [rows, n] = x [rows, n] . matrix [n, n]   fp32 accumulation on the matrix pipe, one rounding to the activation dtype
 * n in {64, 128} (smaller even n, the head-count axis of o_proj: fq_block_quant_* with FQ_OUT_TRANSFORM). y == x is NOT allowed.
 */
int fq_single_trans_f16(const void* x, const void* matrix, int64_t rows, int n, void* y, void* stream);
int fq_single_trans_bf16(const void* x, const void* matrix, int64_t rows, int n, void* y, void* stream);

/* kv_cache.py:54-61 unpack_i4_and_asym_dequantize: y = q * scale - zero, or scale * (q - zero) with FQ_KV_LAC (fp16). */
int fq_kv_dequant_f16(const void* q, const void* param, int64_t rows, int head_dim, int flags, void* y, void* stream);

/*
 * The paged INT4 KV cache of MultiLayerPagedKVCache4Bit (deploy/transformers/kv_cache.py:166-359; the reference binds a
 * vendored FlashInfer: kernels/flashinfer.cu:9-96, include/flashinfer/{page,decode,quantization}.cuh).
 *   kv_data  [pages, num_layers, 2 (k, v), num_heads, page_size, head_dim/2] uint8, low nibble = even feature
 *   kv_param [pages, num_layers, 2, num_heads, page_size, 2] fp16 = (scale, zero) per cached row
 *   kv_indptr [batch+1], kv_indices [n_pages], last_page_offset [batch] int32: request b owns the pages
 *   kv_indices[kv_indptr[b] .. kv_indptr[b+1]); its length is (n_pages_b - 1) * page_size + last_page_offset[b]
 *   (the length AFTER the append, as in page.cuh:135-137,176-183).
 * fq_kv_append_i4: k, v [tokens, num_heads, head_dim/2] uint8 and k_param, v_param [tokens, num_heads, 2] fp16 (the
 *   outputs of fq_kv_quant_f16). seqlen_indptr [batch+1] int32: request b appends tokens seqlen_indptr[b] ..
 *   seqlen_indptr[b+1] - 1 at the END of its current length (init_kv_i4, page.cuh:163-214); NULL: one token per
 *   request, tokens == batch (append_kv_i4, page.cuh:118-161). group_size g > 1 (grouped-query attention,
 *   kv_cache.py:286-296, which repeats the tensors first): k, v and their params hold num_heads / g heads and cache head h
 *   receives head h / g — the repeat_interleave happens in the scatter.
 * fq_kv_batch_decode_i4: one query token per request, q and o [batch, num_heads, head_dim] fp16 (decode.cuh:492-683,
 *   no rotary embedding, softmax scale 1/sqrt(head_dim)): o = softmax(q . K^T / sqrt(hd)) . V over the request's cached
 *   rows, K and V de-quantised as n * scale - zero (quantization.cuh:58-80), fp32 arithmetic. head_dim in {64, 128}.
 */
int fq_kv_append_i4(void* kv_data, void* kv_param, const void* kv_indptr, const void* kv_indices,
                    const void* last_page_offset, const void* k, const void* v, const void* k_param, const void* v_param,
                    const void* seqlen_indptr, int64_t tokens, int num_layers, int layer_idx, int num_heads, int page_size,
                    int head_dim, int batch_size, int group_size, void* stream);
/*
 * fq_kv_quant_f16 on the new keys (with trans) AND values of a layer step, written straight into the paged cache — one
 * launch instead of two quantiser launches and the append: k, v [tokens, src_heads, head_dim] fp16 with
 * tokens = batch_size * added (every request appends the same number of tokens, request-major, at the END of its
 * current length, as fq_kv_append_i4 with seqlen_indptr = added * arange); num_heads = src_heads * group_size cache heads,
 * group_size <= 8. clip = {k_max, k_min, v_max, v_min} (sigmoid-ed; used with FQ_KV_LAC) or NULL. Same bits in the cache
 * as fq_kv_quant_f16 x 2 + fq_kv_append_i4.
 */
int fq_kv_quant_append_i4(const void* k, const void* v, const void* trans, int64_t tokens, int src_heads, int head_dim,
                          const float* clip, int flags, void* kv_data, void* kv_param, const void* kv_indptr,
                          const void* kv_indices, const void* last_page_offset, int num_layers, int layer_idx,
                          int num_heads, int page_size, int batch_size, int group_size, void* stream);
int fq_kv_batch_decode_i4(void* o, const void* q, const void* kv_data, const void* kv_param, const void* kv_indptr,
                          const void* kv_indices, const void* last_page_offset, int num_layers, int layer_idx,
                          int num_heads, int page_size, int head_dim, int batch_size, void* stream);
/* The same with the two torch ops either side of it folded in: q_trans [head_dim, head_dim] fp16 or NULL — the query is
 * first multiplied by it (q' = fp16(q . q_trans), kv_cache.py:139-140, trans_matrix_k_inv_t); transpose_out != 0 — o is
 * written as [batch, head_dim, num_heads], the layout the o_proj head transform takes (modeling_llama.py:147-149). */
int fq_kv_batch_decode_i4_ex(void* o, const void* q, const void* q_trans, int transpose_out, const void* kv_data,
                             const void* kv_param, const void* kv_indptr, const void* kv_indices,
                             const void* last_page_offset, int num_layers, int layer_idx, int num_heads, int page_size,
                             int head_dim, int batch_size, void* stream);

/* (round 6, extension) The decode attention over a cache that holds the KV heads ONCE: grouped-query attention without the per-query-head
 * copies the reference's cache class makes (kv_cache.py:286-296 repeats every KV head num_heads / num_kv_heads times at append time). kv_data /
 * kv_param are laid out with num_kv_heads heads (filled by fq_kv_quant_append_i4 / fq_kv_append_* with num_heads = num_kv_heads,
 * group_size = 1); q and o hold num_kv_heads * q_group query heads, query head h reads cache head h / q_group. The same rows, the same
 * arithmetic: o is bit-identical to fq_kv_batch_decode_split on the replicated cache — from 1 / q_group of the cache memory, and with the
 * q_group workgroups of a KV head re-reading rows the memory-side cache still holds (profiles/r06_gqa_cache.txt). fp16_cache, q_trans,
 * transpose_out, seq_hint: as fq_kv_batch_decode_split; workspace: fq_kv_decode_workspace_bytes_gqa(batch, num_kv_heads, q_group, head_dim)
 * bytes (0: this geometry never splits; NULL: no split). At head_dim 128 with q_group 2 or 4 ONE workgroup per (request, KV head) serves its
 * query heads from one pass over the rows (loaded and unpacked once; the q . k MFMA carries all of them in rows that were idle). */
int64_t fq_kv_decode_workspace_bytes_gqa(int batch_size, int num_kv_heads, int q_group, int head_dim);
int fq_kv_batch_decode_gqa(int fp16_cache, void* o, const void* q, const void* q_trans, int transpose_out, const void* kv_data,
                           const void* kv_param, const void* kv_indptr, const void* kv_indices, const void* last_page_offset,
                           int num_layers, int layer_idx, int num_kv_heads, int q_group, int page_size, int head_dim, int batch_size, int seq_hint,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* (round 6, third session) The decode step's append INSIDE the decode launch — deploy/transformers/kv_cache.py:283-359: update() quantises and
 * appends the step's K / V row (fq_kv_quant_append_i4) and the returned closure attends over the cache including it (batch_decode_i4): two
 * dependent launches. fq_kv_decode_append_i4 is both: the lengths (kv_indptr / last_page_offset) already count the new token, k_new / v_new
 * [batch, src_heads, head_dim] fp16 are its keys / values, k_trans_image the K transform as fq_kv_transform_image_f16 wrote it
 * (fq_kv_transform_image_bytes(head_dim) bytes, once per layer; NULL: keys are not transformed — trans "had" / none). The workgroup that owns a
 * request's last row quantises the new row with fq_kv_quant_append_i4's arithmetic (lac off, as the cache's own calls), uses it from LDS and
 * writes it to the cache: the cache contents and o are bit-identical to fq_kv_quant_append_i4 followed by fq_kv_batch_decode_gqa.
 * The cache holds num_kv_heads heads = src_heads x group (group <= 8: the reference's replicated layout has num_kv_heads = query heads and
 * q_group = 1; a shared cache num_kv_heads = src_heads). head_dim 128 and page_size % 16 == 0 only (FQ_EUNSUPPORTED otherwise).
 * Everything else as fq_kv_batch_decode_gqa. */
int64_t fq_kv_transform_image_bytes(int head_dim);
int fq_kv_transform_image_f16(const void* trans, int head_dim, void* image, void* stream);
int fq_kv_decode_append_i4(void* o, const void* q, const void* q_trans, int transpose_out, const void* k_new, const void* v_new,
                           const void* k_trans_image, int src_heads, const void* kv_data, const void* kv_param, const void* kv_indptr,
                           const void* kv_indices, const void* last_page_offset, int num_layers, int layer_idx, int num_kv_heads, int q_group,
                           int page_size, int head_dim, int batch_size, int seq_hint, int read_one_copy, void* workspace, int64_t workspace_bytes, void* stream);

/* The reference's replicated layout READ ONE COPY PER GROUP (round 6, third session). kv_cache.py:286-296 stores every KV head once per query head
 * of its group: `copies` consecutive cache heads with identical rows (fq_kv_quant_append_i4 / fq_kv_append_* with group_size = copies write them
 * so). fq_kv_batch_decode_copies computes the attention of fq_kv_batch_decode_split on such a cache — num_heads query heads = cache heads — but
 * reads the rows of cache head (h / copies) * copies for query head h: the same values from 1 / copies of the bytes — o bit for bit where the
 * launch geometry is the same; from 32 (request, group) pairs on one workgroup serves the group's query heads from one pass over the rows (as
 * fq_kv_batch_decode_gqa): another order of the fp32 additions, as with split launches.
 * The caller asserts the copies ARE identical; a cache filled any other way must use fq_kv_batch_decode_split.
 * workspace: fq_kv_decode_workspace_bytes_gqa(batch, num_heads / copies, copies, head_dim). fq_kv_decode_append_i4(read_one_copy != 0) is the
 * same reading for the launch that also appends (it writes the new row to every copy). */
int fq_kv_batch_decode_copies(int fp16_cache, void* o, const void* q, const void* q_trans, int transpose_out, const void* kv_data,
                              const void* kv_param, const void* kv_indptr, const void* kv_indices, const void* last_page_offset,
                              int num_layers, int layer_idx, int num_heads, int copies, int page_size, int head_dim, int batch_size, int seq_hint,
                              void* workspace, int64_t workspace_bytes, void* stream);

/*
 * The fp16 configuration of the same cache (MultiLayerPagedKVCache4Bit(disable_quant=True), kv_cache.py:177-190;
 * init_kv_f16 / append_kv_f16 / batch_decode_f16, kv_cache.py:107-137): kv_data [pages, num_layers, 2, num_heads, page_size,
 * head_dim] fp16; k, v [tokens, num_heads / group_size, head_dim] fp16; k_param / v_param are scattered as in the INT4
 * configuration (the reference passes ones / zeros) and never read by the decode. Everything else as fq_kv_append_i4 /
 * fq_kv_batch_decode_i4[_ex].
 */
int fq_kv_append_f16(void* kv_data, void* kv_param, const void* kv_indptr, const void* kv_indices,
                     const void* last_page_offset, const void* k, const void* v, const void* k_param, const void* v_param,
                     const void* seqlen_indptr, int64_t tokens, int num_layers, int layer_idx, int num_heads, int page_size,
                     int head_dim, int batch_size, int group_size, void* stream);
int fq_kv_batch_decode_f16(void* o, const void* q, const void* kv_data, const void* kv_param, const void* kv_indptr,
                           const void* kv_indices, const void* last_page_offset, int num_layers, int layer_idx,
                           int num_heads, int page_size, int head_dim, int batch_size, void* stream);
/*
 * (round 5) The same decode attention with a request's rows SPLIT over several workgroups: with few (request, head) pairs — one request
 * of a 32-head model is 32 workgroups on a 256-CU part — the launch above streams the cache through a fraction of the chip (the
 * reference's own sweep, benchmarks/qattention_benchmark.py:150-153, starts at batch size 1; FlashInfer partitions the sequence the same way, decode.cuh's
 * partition-kv path). The library picks the split count from batch_size x num_heads (1 above 128 pairs: then this IS the launch
 * above) and from seq_hint, the caller's idea of the longest request (0: unknown; no split is made shorter than 256 rows); each
 * workgroup leaves its partial softmax state in `workspace`, the last one to arrive merges them. Results equal the unsplit launch up to
 * the order of fp32 additions.
 *   workspace: fq_kv_decode_workspace_bytes(batch_size, num_heads, head_dim) bytes (0: this geometry is never split), 16-byte aligned,
 *   ZEROED once before its first use (the launches leave its counters as they found them), used by one launch at a time and for ONE geometry
 *   (batch_size x num_heads, head_dim) — zero it again before using it for another. NULL: no split.
 *   fp16_cache != 0: the fp16 configuration (kv_data as fq_kv_append_f16 lays it out; kv_param unused).
 */
int64_t fq_kv_decode_workspace_bytes(int batch_size, int num_heads, int head_dim);
int fq_kv_batch_decode_split(int fp16_cache, void* o, const void* q, const void* q_trans, int transpose_out, const void* kv_data,
                             const void* kv_param, const void* kv_indptr, const void* kv_indices, const void* last_page_offset,
                             int num_layers, int layer_idx, int num_heads, int page_size, int head_dim, int batch_size, int seq_hint,
                             void* workspace, int64_t workspace_bytes, void* stream);
int fq_kv_batch_decode_f16_ex(void* o, const void* q, const void* q_trans, int transpose_out, const void* kv_data,
                              const void* kv_param, const void* kv_indptr, const void* kv_indices,
                              const void* last_page_offset, int num_layers, int layer_idx, int num_heads, int page_size,
                              int head_dim, int batch_size, void* stream);

/* q = clamp(rn(x /fp16 scale[row]), -8, 7), two per byte, even column -> low nibble (quant.cu:13-47). */
int fq_sym_quant_f16(const void* x, const void* scale, int64_t rows, int cols, void* q, void* stream);

/* x = scale_row[r] * scale_col[c] * half(q/10) * 10   (quant.cu:66-85) */
int fq_sym_dequant_i32_f16(const void* q, const void* scale_row, const void* scale_col,
                           int64_t rows, int cols, void* x, void* stream);

const char* fq_last_error(void);
int fq_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FQHIP_H */
