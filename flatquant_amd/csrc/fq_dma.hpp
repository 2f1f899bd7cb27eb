// fq_dma.hpp — HBM -> LDS token staging by LDS-DMA for the wave-per-token kernels (fq_kron_wave.hip, fq_block.hip).
// (fq_kron64.hip has its own hand-scheduled copy of the same scheme.)
#pragma once
#include "fq_common.hpp"

typedef __attribute__((address_space(3))) void lds_void;

// chunk swizzle of row r inside the token buffer (16-byte chunks, CPR per row): 8 consecutive rows must spread over
// the 16 chunk-aligned bank groups. CPR = 16: rows alias -> XOR the row's low bits; CPR = 8: pairs of rows alias;
// CPR = 14 (N = 112) and CPR = 10 (N = 80): the row pitch already rotates (14 r, 10 r mod 16 are eight distinct even slots).
template <int CPR>
__device__ __forceinline__ int swz(int r) {
    return CPR == 16 ? (r & 15) : CPR == 8 ? ((r >> 1) & 7) : 0;
}

// DMA of one token (n_dma instructions of 1 KB) into the wave's buffer. Instruction i moves the LDS slots
// [64 i, 64 i + 64) (lane-linear); lane l fetches the global chunk that the swizzle maps to its slot. The per-lane
// byte offset inside the instruction's KB only depends on i & 3 (CPR = 16), i & 1 (CPR = 8) or not at all, so the
// caller precomputes four of them (dma_offsets); four instructions share one M0 / base pair through the
// instruction offset field, which advances the global AND the LDS address (see fq_kron64.hip).
template <int CPR>
__device__ __forceinline__ void dma_offsets(int lane, unsigned (&voff)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * 64 + lane;                    // LDS slot
        const int r = q / CPR, pch = q - r * CPR;
        voff[i] = (unsigned)((r * CPR + (pch ^ swz<CPR>(r))) * 16 - i * 1024);  // relative to this instruction's KB
    }
}
template <int CPR>
__device__ __forceinline__ void dma_token(const f16* __restrict__ x, int64_t tok, int64_t tok_bytes, int n_dma,
                                          unsigned lds_base, const unsigned (&voff)[4]) {
    static_assert(CPR == 16 || CPR == 8 || CPR == 14 || CPR == 10, "offset pattern must repeat every 4 instructions");
    const unsigned char* base = reinterpret_cast<const unsigned char*>(x) + tok * tok_bytes;  // wave-uniform
    const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)base);
    const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((size_t)base >> 32));
    const unsigned long long sb = (unsigned long long)lo32 | ((unsigned long long)hi32 << 32);
    int g = 0;
    for (; g + 4 <= n_dma; g += 4) {
        unsigned keep;
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %6\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %5 nt\n\t"
            "global_load_lds_dwordx4 %2, %5 offset:1024 nt\n\t"
            "global_load_lds_dwordx4 %3, %5 offset:2048 nt\n\t"
            "global_load_lds_dwordx4 %4, %5 offset:3072 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(sb + (unsigned long long)g * 1024),
              "s"(lds_base + (unsigned)g * 1024)
            : "memory");
    }
    for (int j = 0; g + j < n_dma; ++j) {  // tail (n_dma % 4 instructions)
        unsigned keep;
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(j == 0 ? voff[0] : j == 1 ? voff[1] : voff[2]), "s"(sb + (unsigned long long)(g + j) * 1024),
              "s"(lds_base + (unsigned)(g + j) * 1024)
            : "memory");
    }
}

// A wave's share of a token that several waves stage together (fq_kron_trio.hip): instructions [i0, i0 + n) of the token,
// i.e. the KBs [i0, i0 + n) of its LDS image. `src` / `lds_base` already point at KB i0; rv[j] is the per-lane offset
// of instruction i0 + j (dma_span_offsets), which repeats every four instructions.
template <int CPR>
__device__ __forceinline__ void dma_span_offsets(int lane, int i0, unsigned (&rv)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = i0 + j, q = i * 64 + lane;
        const int r = q / CPR, pch = q - r * CPR;
        rv[j] = (unsigned)((r * CPR + (pch ^ swz<CPR>(r))) * 16 - i * 1024);
    }
}
__device__ __forceinline__ void dma_span(const unsigned char* src, int n, unsigned lds_base, const unsigned (&rv)[4]) {
    const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)src);
    const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((size_t)src >> 32));
    const unsigned long long sb = (unsigned long long)lo32 | ((unsigned long long)hi32 << 32);
    int g = 0;
    for (; g + 4 <= n; g += 4) {
        unsigned keep;
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %6\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %5 nt\n\t"
            "global_load_lds_dwordx4 %2, %5 offset:1024 nt\n\t"
            "global_load_lds_dwordx4 %3, %5 offset:2048 nt\n\t"
            "global_load_lds_dwordx4 %4, %5 offset:3072 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(rv[0]), "v"(rv[1]), "v"(rv[2]), "v"(rv[3]), "s"(sb + (unsigned long long)g * 1024),
              "s"(lds_base + (unsigned)g * 1024)
            : "memory");
    }
    for (int j = 0; g + j < n; ++j) {  // tail (n % 4 instructions)
        unsigned keep;
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(j == 0 ? rv[0] : j == 1 ? rv[1] : rv[2]), "s"(sb + (unsigned long long)(g + j) * 1024),
              "s"(lds_base + (unsigned)(g + j) * 1024)
            : "memory");
    }
}
