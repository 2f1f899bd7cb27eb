/*
 * fqprobe.h — C ABI of libfqprobe.so: MEASUREMENT / TEST INFRASTRUCTURE for the MI355X build, kept OUT of the product
 * library and of include/fqhip.h. Built from flatquant_amd/csrc/probe/fq_probe.hip by the same Makefile; loaded only by
 * bench.py (the streaming floor it quotes next to the 8 TB/s spec peak), tools/ and tests/ (flatquant_amd/_probe.py).
 * Same conventions as fqhip.h: device pointers owned by the caller, `stream` = hipStream_t or NULL, 0 on success.
 */
#ifndef FQPROBE_H
#define FQPROBE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* MFMA numerics probe (the oracle's accumulation model): D[32,32] = A[32,16] . B[16,32] + C with ONE
 * v_mfma_f32_32x32x16_f16. All row-major, A / B fp16, C / D fp32. */
int fq_probe_mfma_32x32x16_f16(const void* A, const void* B, const void* C, void* D, void* stream);

/* HBM-floor probe: streams exactly the bytes of the d = 4096 fused kernel (8192 B read, 2048 B + 2 B written per token) with
 * fully coalesced 16-byte accesses and no arithmetic. x [rows, 4096] fp16, q [rows, 2048] bytes, s [rows] fp16.
 * waves_per_simd (1..8) sizes the persistent grid. */
int fq_probe_stream_4096(const void* x, int64_t rows, void* q, void* s, int waves_per_simd, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FQPROBE_H */
