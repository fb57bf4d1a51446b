// Register-level helpers of the one-wave-per-SIMD GEMM kernels (gemm4.hip: 128 x 128 outputs per wave; tools/probes/gemm4h_experiment.hip: the same block as two
// 64-row halves whose epilogue runs under the other half's K loop): the accumulator file as LITERAL AGPRs behind inline-asm MFMAs, the
// clobber list that keeps the compiler out of it, accumulator reads, the LDS-DMA piece.  (Moved out of gemm4.hip unchanged in round 6.)
#pragma once
#include <utility>
#include "gemm_shared.h"

namespace {

template <typename F, int... I>
__device__ __forceinline__ void g4_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void g4_for(F&& f) {
    g4_for_impl(f, std::make_integer_sequence<int, N>{});
}

#define G4_A16(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
// EVERY accumulator register, as a clobber list.  Each asm statement below carries it: (i) the kernel descriptor allocates all 256, and
// (ii) no compiler value can live in an AGPR across any of these statements -- the compiler, which believes the AGPRs are free, would
// otherwise park spilled VGPRs there (seen: the LayerNorm-consuming instantiation wrote 30 of them over live accumulators).  With the
// list a register shortage becomes an ordinary scratch spill, which the build audit (`make check4`) rejects.
#define G4_ALL_ACC                                                                                                                              \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", G4_A16(1), G4_A16(2), G4_A16(3), G4_A16(4), G4_A16(5), G4_A16(6), G4_A16(7),    \
        G4_A16(8), G4_A16(9), G4_A16(10), G4_A16(11), G4_A16(12), G4_A16(13), G4_A16(14), G4_A16(15), G4_A16(16), G4_A16(17), G4_A16(18),        \
        G4_A16(19), G4_A16(20), G4_A16(21), G4_A16(22), G4_A16(23), G4_A16(24), "a250", "a251", "a252", "a253", "a254", "a255"

// acc block I (a[4I .. 4I+3]) (+)= W-fragment x A-fragment; ZERO: C = 0 (the tile's first k-step)
// (CLOB: this statement carries the clobber list -- the first MFMA of every quarter and the first read of every epilogue chunk do; on
// all ~800 statements the list cost a minute of compile time per instantiation)
// HF: f16 operands (MAGE_F16) -- the other 16-bit opcode, same rate, same operand and accumulator layout
#define G4_MFMA_STMT(OP)                                                                                                                                   \
    do {                                                                                                                                                   \
        if constexpr (ZERO && CLOB) asm volatile(OP " a[%c2:%c3], %0, %1, 0" ::"v"(w), "v"(x), "i"(4 * I), "i"(4 * I + 3) : G4_ALL_ACC);                   \
        else if constexpr (ZERO) asm volatile(OP " a[%c2:%c3], %0, %1, 0" ::"v"(w), "v"(x), "i"(4 * I), "i"(4 * I + 3));                                  \
        else if constexpr (CLOB) asm volatile(OP " a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(x), "i"(4 * I), "i"(4 * I + 3) : G4_ALL_ACC);             \
        else asm volatile(OP " a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(x), "i"(4 * I), "i"(4 * I + 3));                                             \
    } while (0)
template <int I, bool ZERO, bool CLOB, bool HF = false>
__device__ __forceinline__ void g4_mfma(const u32x4& w, const u32x4& x) {
    if constexpr (HF) G4_MFMA_STMT("v_mfma_f32_16x16x32_f16");
    else G4_MFMA_STMT("v_mfma_f32_16x16x32_bf16");
}
template <int I, bool CLOB = false>
__device__ __forceinline__ f32x4 g4_acc_read() {
    f32x4 v;
    if constexpr (CLOB)
        asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7"
                     : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
                     : "i"(4 * I), "i"(4 * I + 1), "i"(4 * I + 2), "i"(4 * I + 3)
                     : G4_ALL_ACC);
    else
        asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7"
                     : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
                     : "i"(4 * I), "i"(4 * I + 1), "i"(4 * I + 2), "i"(4 * I + 3));
    return v;
}
template <int I>
__device__ __forceinline__ void g4_acc_write(const f32x4& v) {
    asm volatile("v_accvgpr_write_b32 a%c4, %0\n\tv_accvgpr_write_b32 a%c5, %1\n\tv_accvgpr_write_b32 a%c6, %2\n\tv_accvgpr_write_b32 a%c7, %3" ::"v"(v[0]),
                 "v"(v[1]), "v"(v[2]), "v"(v[3]), "i"(4 * I), "i"(4 * I + 1), "i"(4 * I + 2), "i"(4 * I + 3)
                 : G4_ALL_ACC);
}
__device__ __forceinline__ void g4_claim_accumulators() { asm volatile("" ::: G4_ALL_ACC); }
// LDS-DMA of unit U of a group of 8: 8 rows x 128 B from (SGPR base + 32-bit lane offset + (U - 4) KiB) to LDS (M0 + (U - 4) KiB + lane * 16)
template <int U>
__device__ __forceinline__ void g4_dma(unsigned voff, const char* base) {
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%c2" ::"v"(voff), "s"(base), "i"((U - 4) * 1024) : "memory");
}


}  // namespace
