// gemm4_kernel: the bf16 256x256 plain GEMM with ONE wave per SIMD (4 waves of 128x128 outputs each, 256 accumulator registers per
// lane, launch_bounds(256): the 512-register budget) -- the decoder's Linear layers at M >= 2 tiles per CU (see include/mage_hip.h,
// mage_gemm; dispatch in gemm.hip).
//
// Why another tile shape.  The 8-wave kernels (gemm.hip) give each wave 128x64 outputs: per 64-wide K slab a wave reads 28 (8-phase)
// operand fragments of 1 KiB for 64 MFMAs, 224 KiB of LDS reads per slab per CU beside the slab's own 64 KiB of LDS-DMA writes, and the
// 8-phase schedule pays two barriers per 16-MFMA section: 3.1-3.2 k cycles per slab against 2.05 k of matrix-pipe time
// (profiles/r03_gemm_tile_phases.txt).  With 128x128 per wave a slab is 32 fragment reads for 128 MFMAs per wave -- 128 KiB per CU, half
// the fragment traffic per MAC -- there is no partner wave to hand the SIMD to, so no intra-slab barrier at all: ONE barrier per slab,
// and the wave's own instruction stream interleaves its ds_reads and LDS-DMA issues between its MFMAs (MI355X_MICROARCH.md: one wave
// per SIMD hides <= 5 single-issue instructions per 32-cycle MFMA gap; this loop needs < 1 per 16-cycle gap).
//
// The accumulators are the WHOLE accumulator file (a0..a255).  hipcc's register allocator cannot hold a 256-register value set in a
// 256-register class: given the MFMAs as builtins it split the accumulators between AGPRs and VGPRs and copied them back and forth
// inside the K loop (3800 v_accvgpr_write, 100 spilled VGPRs).  So the MFMAs are inline asm on LITERAL accumulator registers
// (cdna_hip_programming.md 5.7): the compiler never sees an accumulator value; it allocates only the operand fragments, the loader's
// offsets and the epilogue's temporaries (< 256 VGPRs, no spill: `make check4` audits the .s for spills and for compiler-made
// v_accvgpr_* outside the asm blocks).  The first k-step of a tile multiplies into C = 0 (no zeroing pass); the epilogue reads the
// accumulators back 64 registers at a time (v_accvgpr_read) into the array epilogue_lean takes.
//
// LDS: 2 stages x (256 A rows + 256 W rows) x 128 B = 128 KiB, the lane-linear XOR-swizzled image of gemm.hip (physical 16-byte chunk p
// of row r holds logical chunk p ^ ((r >> 1) & 7); applied on the DMA source address and on the ds_read_b128), + 4 KiB of epilogue
// staging per wave: 144 KiB.
//
// Schedule of slab j (stage s = j & 1), four QUARTERS of 32 MFMAs = (k-half t, row half mh), loads placed BETWEEN the MFMAs in source
// order and pinned there (sched_barrier):
//     Q0 (t0, rows 0-63)    reads: A(t0, rows 64-127)                                   DMA: W pieces of slab j+1 -> stage s^1
//     Q1 (t0, rows 64-127)  reads: W(t1) x 8, A(t1, rows 0-63)
//     Q2 (t1, rows 0-63)    reads: A(t1, rows 64-127);  then  s_waitcnt vmcnt(0) lgkmcnt(0);  s_barrier
//     Q3 (t1, rows 64-127)  reads from stage s^1: W'(t0) x 8, A'(t0, rows 0-63)          DMA: A pieces of slab j+2 -> stage s
// The barrier sits at the 3/4 point: by then this wave has issued (and waited for) its last read of stage s, and its share of slab
// j+1 has landed; past it stage s is free for slab j+2 and slab j+1 is visible, while the wave still holds the operands of 32 MFMAs
// in registers -- the matrix pipe restarts without an LDS round trip.  A pieces (streamed from HBM) are requested 1.75 slabs ahead
// of their first read, W pieces (L2-resident) 0.75.  The slab sequence runs across tile boundaries: Q3 of a tile's last slab already
// fetches the next tile's first fragments, the epilogue runs, the next tile's Q0 starts from registers.
//
// Arithmetic: v_mfma_f32_16x16x32_bf16, operands swapped (A-operand = W rows) and k ascending in steps of 32 exactly as in gemm.hip's
// three kernels, the SAME epilogue_lean on the same accumulator layout: every output element gets the same bits from all four kernels
// (incremental == full loop, B = 1 == row of a batch stay exact).
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "gemm_shared.h"
#include "gemm4_regs.h"

namespace {

struct Gemm4Args {
    const void* A;
    const void* W;
    void* Y;
    const float* bias;
    const float* ln_stats;
    const float* ln_colsum;
    const void* residual;
    float* ln_part;
    void* y2;
    int M, N, K, lda, ldw, ldy, ldr, ldy2, y_dtype, a_off, y_off;
    int ntiles_n, ntiles;
    int stagger_groups, stagger_sleeps;
};

#ifndef MAGE4_ABL
#define MAGE4_ABL 0      // tuning builds, bit flags: 1 = no epilogue (K loop only), 2 = no LDS-DMA in the loop, 4 = no fragment reads in the loop,
                         // 8 = LDS-DMA always from the first slab of the first tile (cache-resident source), 16 = no vmcnt wait at the barrier,
                         // 32 = every tile's output goes to one fixed window per workgroup (cache-resident stores), 64 = epilogue without stores
#endif

#ifdef MAGE4_STAMP
// tuning build: shader-clock stamps of every wave per (workgroup, tile < 32): [0] tile start, [1] K loop done, [2] epilogue done, [3] 100 MHz wall clock
// at [1], [4] 100 MHz wall clock at [0] (shader cycles per wall microsecond over the K loop = the clock the chip actually holds)
__device__ unsigned long long g4_stamps[256 * 32 * 4 * 8];
#define G4_STAMP(it, p)                                                                                                   \
    do {                                                                                                                  \
        if (lane == 0 && (it) < 32 && blockIdx.x < 256) {                                                                 \
            g4_stamps[((blockIdx.x * 32 + (it)) * 4 + wave) * 8 + (p)] = __builtin_readcyclecounter();                    \
            if ((p) == 1) g4_stamps[((blockIdx.x * 32 + (it)) * 4 + wave) * 8 + 3] = __builtin_amdgcn_s_memrealtime();    \
            if ((p) == 0) g4_stamps[((blockIdx.x * 32 + (it)) * 4 + wave) * 8 + 4] = __builtin_amdgcn_s_memrealtime();    \
        }                                                                                                                 \
    } while (0)
#else
#define G4_STAMP(it, p)
#endif

constexpr int G4_STAGE = 65536, G4_WOFF = 32768, G4_RING = 2 * G4_STAGE, G4_LDS = G4_RING + 4 * 4096;      // 144 KiB

#ifdef MAGE4_MFMA32
// Tuning build (tools/probes/gemm4_probe.hip -DMAGE4_MFMA32 -DMAGE4_ABL=1): the K loop on v_mfma_f32_32x32x16_bf16 -- 16 blocks of 32 x 32 outputs
// (16 accumulator registers each) instead of 64 of 16 x 16: half the MFMA issues and half the operand-register reads per FLOP.  K loop only:
// the epilogue is written for the 16 x 16 accumulator layout, so this build answers one question -- does the same slab of work draw less
// power (hold a higher clock) in the other MFMA shape -- and nothing else.
template <int B, bool ZERO, bool CLOB>
__device__ __forceinline__ void g4_mfma32(const u32x4& w, const u32x4& x) {
    if constexpr (ZERO && CLOB) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(w), "v"(x), "i"(16 * B), "i"(16 * B + 15) : G4_ALL_ACC);
    else if constexpr (ZERO) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(w), "v"(x), "i"(16 * B), "i"(16 * B + 15));
    else if constexpr (CLOB) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(x), "i"(16 * B), "i"(16 * B + 15) : G4_ALL_ACC);
    else asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(x), "i"(16 * B), "i"(16 * B + 15));
}
#endif
template <int ACT, int EK, int LN, bool RB, bool HF = false>
__global__ __launch_bounds__(256) void gemm4_kernel(const Gemm4Args g) {
    static_assert(EK != EK_GENERAL, "lean epilogue kinds");
    static_assert(!HF || (LN != LN_DUAL && LN != LN_GELUBWD), "f16 form: the generation path's epilogues");
    typedef std::conditional_t<HF, f16_t, unsigned short> H16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    g4_claim_accumulators();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // persistent tile schedule of gemm.hip: workgroup b runs on XCD b % 8 (speed only); each XCD owns a contiguous chunk of the tile list
    const int nwg8 = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int q8 = g.ntiles >> 3, r8 = g.ntiles & 7;
    const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int chunk1 = chunk0 + q8 + (xcd < r8 ? 1 : 0);
    const int nk = g.K >> 6;                           // K % 128 == 0, K >= 256 (host): FIRST and LAST slabs are peeled, the rest runs in pairs
    int c_tile = chunk0 + li;
    if (c_tile >= chunk1) return;
    const int last_tile = c_tile + ((chunk1 - 1 - c_tile) / nwg8) * nwg8;      // this workgroup's last tile (the cursors stop there)

    // ---- loader: per slab this wave moves 8 A units and 8 W units (unit = 8 rows x 128 B = one wave-wide LDS-DMA) of its 64-row strips.
    // global_load_lds_dwordx4 with an SGPR base and a 32-bit lane offset (no vector instruction per piece).  Inline asm, because the builtin
    // takes one 64-bit pointer per lane and an M0 write per piece: here M0 (the LDS destination base) is written ONCE per group of 8
    // units and the instruction's immediate offset (u - 4) KiB, which the hardware adds to the LDS address AND to the global address,
    // selects the unit; the lane offsets carry the opposite shift (and + 4 KiB, taken off the base pointers, to stay non-negative).
    // Nothing else in this kernel reads M0 (audited in the .s); the statements have no VGPR result, their completion is the explicit
    // vmcnt wait in front of each slab's barrier.
    const int lr = lane >> 3, lp = lane & 7;
    unsigned voffA[8], voffW[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int r = u * 8 + lr;
        const int ch = lp ^ ((r >> 1) & 7);
        voffA[u] = (unsigned)(r * g.lda * 2 + ch * 16 + 8192 - u * 1024);
        voffW[u] = (unsigned)(r * g.ldw * 2 + ch * 16 + 8192 - u * 1024);
    }
    auto a_base = [&](int tile) { return (const char*)g.A + ((long)((tile / g.ntiles_n) * 256 + wave * 64) + g.a_off) * g.lda * 2 - 4096; };
    auto w_base = [&](int tile) { return (const char*)g.W + (long)((tile % g.ntiles_n) * 256 + wave * 64) * g.ldw * 2 - 4096; };
    // two cursors: A pieces run 2 slabs ahead of the compute cursor, W pieces 1; at a tile's end they jump to the next tile's bases
    const char* a_src = a_base(c_tile);
    const char* w_src = w_base(c_tile);
    const char* a_next = a_src;
    const char* w_next = w_src;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 8192 + 4096;     // (8 units of 1 KiB per wave and operand)
    auto dma_m0 = [&](unsigned lds_group) __attribute__((always_inline)) {       // lds_group: stage * G4_STAGE (+ G4_WOFF)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds0 + lds_group) : "memory");      // (m0 cannot be named as a clobber: the compiler treats it as reserved and warns; check4 audits its writers)
    };

    // ---- compute state: wave (wm, wn) owns rows [wm*128, +128) x columns [wn*128, +128) of the tile
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, grp = lane >> 4;
    const int rsw = (l15 >> 1) & 7;
    const int pc[2] = {((grp + 0) ^ rsw) * 16, ((grp + 4) ^ rsw) * 16};
    const int x_off = (wm * 128 + l15) * 128;
    const int w_off = G4_WOFF + (wn * 128 + l15) * 128;
    u32x4 wf[2][8], xf[2][4];
    // accumulator block of (column half h, 16-row tile mt, 16-column block j): columns h*64 + j*16 + grp*4 + {0..3} of row mt*16 + l15
    // index = (h*8 + mt)*4 + j
    auto rd_w1 = [&](int buf, int stage, int t, int nt) __attribute__((always_inline)) {
        wf[buf][nt] = *(const u32x4*)(smem + stage * G4_STAGE + w_off + nt * 2048 + pc[t]);
    };
    auto rd_x1 = [&](int buf, int stage, int t, int mt) __attribute__((always_inline)) {
        xf[buf][mt & 3] = *(const u32x4*)(smem + stage * G4_STAGE + x_off + mt * 2048 + pc[t]);
    };

#ifdef MAGE4_MFMA32
    // 32 x 32 x 16: lane (l31 = lane & 31, hi = lane >> 5) holds 8 consecutive k (16 bytes) of fragment row l31: k-step s of a slab is logical
    // chunks 2s + hi of the 128-byte row
    const int l31 = lane & 31, hi5 = lane >> 5;
    const int rsw32 = (l31 >> 1) & 7;
    const int x32_off = (wm * 128 + l31) * 128, w32_off = G4_WOFF + (wn * 128 + l31) * 128;
    u32x4 wq[2][4], xq[2][4];
    auto rd_w32 = [&](int buf, int stage, int ks, int nb) __attribute__((always_inline)) {
        wq[buf][nb] = *(const u32x4*)(smem + stage * G4_STAGE + w32_off + nb * 4096 + (((2 * ks + hi5) ^ rsw32) << 4));
    };
    auto rd_x32 = [&](int buf, int stage, int ks, int mb) __attribute__((always_inline)) {
        xq[buf][mb] = *(const u32x4*)(smem + stage * G4_STAGE + x32_off + mb * 4096 + (((2 * ks + hi5) ^ rsw32) << 4));
    };
#endif
    if (g.stagger_groups > 1) {
        for (int w = (li % g.stagger_groups) * g.stagger_sleeps; w > 0; --w) __builtin_amdgcn_s_sleep(16);
    }
    // prologue: slab 0 complete in stage 0, slab 1 on its way to stage 1, slab 0's first fragments in registers
    dma_m0(0);
    g4_for<8>([&](auto u_) { g4_dma<decltype(u_)::value>(voffA[decltype(u_)::value], a_src); });
    dma_m0(G4_WOFF);
    g4_for<8>([&](auto u_) { g4_dma<decltype(u_)::value>(voffW[decltype(u_)::value], w_src); });
    a_src += 128;
    w_src += 128;
    dma_m0(G4_STAGE);
    g4_for<8>([&](auto u_) { g4_dma<decltype(u_)::value>(voffA[decltype(u_)::value], a_src); });
    dma_m0(G4_STAGE + G4_WOFF);
    g4_for<8>([&](auto u_) { g4_dma<decltype(u_)::value>(voffW[decltype(u_)::value], w_src); });
    a_src += 128;
    w_src += 128;
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): slabs 0 AND 1 (a tile's first slab waits with a count that assumes the previous
                                                       // tile's epilogue traffic behind slab 1's pieces; there is none in front of the first tile)
    ring_barrier();
#ifdef MAGE4_MFMA32
#pragma unroll
    for (int b = 0; b < 4; ++b) { rd_w32(0, 0, 0, b); rd_x32(0, 0, 0, b); }
#else
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) rd_w1(0, 0, 0, nt);
#pragma unroll
    for (int m = 0; m < 4; ++m) rd_x1(0, 0, 0, m);
#endif

    [[maybe_unused]] int it = 0;
    for (; c_tile < chunk1; c_tile += nwg8, ++it) {
        G4_STAMP(it, 0);
        const int tm = c_tile / g.ntiles_n, tn = c_tile - tm * g.ntiles_n;
        MAGE_DASSERT(c_tile >= 0 && c_tile < g.ntiles && tm * 256 < g.M && tn * 256 < g.N);
        const int m0 = ((MAGE4_ABL & 32) ? (int)blockIdx.x : tm) * 256 + wm * 128, n0 = ((MAGE4_ABL & 32) ? 0 : tn) * 256 + wn * 128;
        {
            const int nt_tile = c_tile + nwg8 > last_tile ? last_tile : c_tile + nwg8;     // past the end: re-fetch the last tile (never read)
            a_next = a_base(nt_tile);
            w_next = w_base(nt_tile);
        }
        f32x4 biasm[2][4];
        [[maybe_unused]] float ln_mean[8], ln_rstd[8];
        [[maybe_unused]] f32x4 lns[2][4];
        // One slab.  j = its index in the tile (runtime; only the cursors' tile jumps depend on it), S = its stage.
        // LDS-DMA issue points: W(j+1) in Q0 and A(j+2) in Q3, one instruction per four MFMAs.  A tile's FIRST slab issues no W: the
        // previous tile's LAST slab sent W of this tile's slab 1 together with its A in Q3, BEFORE the epilogue's stores -- memory
        // operations retire in order, so a wait for pieces issued after the stores would also wait for every store acknowledgement
        // (the 8-wave kernels' "first-slab wait", 3.3 k cycles per tile); with both operands of slab 1 older than the stores the FIRST
        // slab waits with a count that leaves the stores in flight.
        auto slab = [&](auto S_, auto FIRST_, auto LAST_, int j) __attribute__((always_inline)) {
            constexpr int S = decltype(S_)::value;
            constexpr bool FIRST = decltype(FIRST_)::value, LAST = decltype(LAST_)::value;
            // ---- Q0: (t0, rows 0-63) from wf[0], xf[0]
            g4_for<32>([&](auto i_) {
                constexpr int i = decltype(i_)::value, m = i >> 3, nt = i & 7;
                g4_mfma<((nt >> 2) * 8 + m) * 4 + (nt & 3), FIRST, i == 0, HF>(wf[0][nt], xf[0][m]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(MAGE4_ABL & 4) && i < 4) {
                    rd_x1(1, S, 0, 4 + i);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (!(MAGE4_ABL & 2) && !FIRST && i == 4) {
                    dma_m0((S ^ 1) * G4_STAGE + G4_WOFF);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (!(MAGE4_ABL & 2) && !FIRST && (i & 3) == 1 && i > 4) {
                    g4_dma<(i >> 2) - 1>(voffW[(i >> 2) - 1], w_src);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            if constexpr (!(MAGE4_ABL & 2) && !FIRST) {
                g4_dma<7>(voffW[7], w_src);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!(MAGE4_ABL & 8) && !FIRST) w_src = (j == nk - 2) ? w_next : w_src + 128;
            // ---- Q1: (t0, rows 64-127) from wf[0], xf[1]
            g4_for<32>([&](auto i_) {
                constexpr int i = decltype(i_)::value, m = i >> 3, nt = i & 7;
                g4_mfma<((nt >> 2) * 8 + 4 + m) * 4 + (nt & 3), FIRST, i == 0, HF>(wf[0][nt], xf[1][m]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(MAGE4_ABL & 4) && i < 12) {
                    if constexpr (i < 4) rd_x1(0, S, 1, i);
                    else rd_w1(1, S, 1, i - 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            // ---- Q2: (t1, rows 0-63) from wf[1], xf[0]
            if constexpr (LAST) {                      // the epilogue's vectors, late: their registers are free during the K loop
#pragma unroll
                for (int b = 0; b < 4; ++b) {          // column half 0 (half 1: requested in the epilogue, under half 0's arithmetic)
                    const int n = n0 + b * 16 + grp * 4;
                    biasm[0][b] = *(const f32x4*)(g.bias + n);                         // bias != null (host)
                    if constexpr (LN == LN_CONSUME) lns[0][b] = *(const f32x4*)(g.ln_colsum + n);
                }
                if constexpr (LN == LN_CONSUME) {
#pragma unroll
                    for (int a = 0; a < 8; ++a) {
                        const float2 st = *(const float2*)(g.ln_stats + 2 * (long)(m0 + a * 16 + l15));
                        ln_mean[a] = st.x;
                        ln_rstd[a] = st.y;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            g4_for<32>([&](auto i_) {
                constexpr int i = decltype(i_)::value, m = i >> 3, nt = i & 7;
                g4_mfma<((nt >> 2) * 8 + m) * 4 + (nt & 3), false, i == 0, HF>(wf[1][nt], xf[0][m]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(MAGE4_ABL & 4) && i < 4) {
                    rd_x1(1, S, 1, 4 + i);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            // my reads of this stage are done and my share of the next slab has landed.  FIRST: the next slab's 16 pieces are older than
            // the previous tile's epilogue traffic (4 or 8 vector loads and >= 32 stores per wave), which may stay in flight (the count waited
            // for is four below that traffic: a margin against any reordering of the epilogue's last stores)
            if constexpr (MAGE4_ABL & 16) __builtin_amdgcn_s_waitcnt(0xC07F);
            else if constexpr (FIRST) __builtin_amdgcn_s_waitcnt(LN == LN_CONSUME ? 0x8074 : 0x8070);       // vmcnt(36 | 32) lgkmcnt(0): four below the count
            else __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
            ring_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- Q3: (t1, rows 64-127) from wf[1], xf[1]; next slab's first fragments into wf[0], xf[0]
            g4_for<32>([&](auto i_) {
                constexpr int i = decltype(i_)::value, m = i >> 3, nt = i & 7;
                g4_mfma<((nt >> 2) * 8 + 4 + m) * 4 + (nt & 3), false, i == 0, HF>(wf[1][nt], xf[1][m]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(MAGE4_ABL & 4) && i < 12 && !LAST) {     // (a tile's last slab: after the epilogue, whose registers these would occupy)
                    if constexpr (i < 4) rd_x1(0, S ^ 1, 0, i);
                    else rd_w1(0, S ^ 1, 0, i - 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (!(MAGE4_ABL & 2) && !LAST) {               // A(j+2): 8 pieces
                    if constexpr (i == 2) dma_m0(S * G4_STAGE);
                    if constexpr ((i & 3) == 3) g4_dma<(i >> 2)>(voffA[i >> 2], a_src);
                    if constexpr (i == 2 || (i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (!(MAGE4_ABL & 2) && LAST) {                // A and W of the next tile's slab 1: 16 pieces
                    if constexpr (i == 0) dma_m0(S * G4_STAGE);
                    if constexpr (i == 16) dma_m0(S * G4_STAGE + G4_WOFF);
                    if constexpr ((i & 1) == 1 && i < 16) g4_dma<(i >> 1)>(voffA[i >> 1], a_src);
                    if constexpr ((i & 1) == 1 && i > 16) g4_dma<((i - 16) >> 1)>(voffW[(i - 16) >> 1], w_src);
                    if constexpr (i == 0 || i == 16 || (i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
                }
            });
            if constexpr (!(MAGE4_ABL & 8)) a_src = (j == nk - 3) ? a_next : a_src + 128;
            if constexpr (!(MAGE4_ABL & 8) && LAST) w_src += 128;
        };
#ifdef MAGE4_MFMA32
        // the same slab on 32 x 32 x 16 MFMAs: quarter q = k-step q (16 of the slab's 64 k), 16 MFMAs over the 4 x 4 blocks; the next k-step's 8
        // fragments are read during the quarter, the DMA issue points, the barrier and the waits are the 16 x 16 schedule's
        auto slab32 = [&](auto S_, auto FIRST_, auto LAST_, int j) __attribute__((always_inline)) {
            constexpr int S = decltype(S_)::value;
            constexpr bool FIRST = decltype(FIRST_)::value, LAST = decltype(LAST_)::value;
            auto quarter = [&](auto Q_) __attribute__((always_inline)) {
                constexpr int Q = decltype(Q_)::value, buf = Q & 1;
                g4_for<16>([&](auto i_) {
                    constexpr int i = decltype(i_)::value, mb = i >> 2, nb = i & 3;
                    g4_mfma32<mb * 4 + nb, FIRST && Q == 0, i == 0>(wq[buf][nb], xq[buf][mb]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!(MAGE4_ABL & 4) && i < 8 && !(Q == 3 && LAST)) {        // next k-step's fragments (Q3: the next slab's first, other stage)
                        constexpr int st = Q == 3 ? (S ^ 1) : S, ks = (Q + 1) & 3;
                        if constexpr (i < 4) rd_x32(buf ^ 1, st, ks, i);
                        else rd_w32(buf ^ 1, st, ks, i - 4);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (Q == 0 && !(MAGE4_ABL & 2) && !FIRST) {                  // W(j+1): 8 pieces
                        if constexpr (i == 4) dma_m0((S ^ 1) * G4_STAGE + G4_WOFF);
                        if constexpr (i >= 5 && i < 13) g4_dma<i - 5>(voffW[i - 5], w_src);
                        if constexpr (i >= 4 && i < 13) __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (Q == 3 && !(MAGE4_ABL & 2) && !LAST) {                   // A(j+2): 8 pieces
                        if constexpr (i == 1) dma_m0(S * G4_STAGE);
                        if constexpr (i >= 2 && i < 10) g4_dma<i - 2>(voffA[i - 2], a_src);
                        if constexpr (i >= 1 && i < 10) __builtin_amdgcn_sched_barrier(0);
                    }
                });
            };
            quarter(std::integral_constant<int, 0>{});
            if constexpr (!(MAGE4_ABL & 8) && !FIRST) w_src = (j == nk - 2) ? w_next : w_src + 128;
            quarter(std::integral_constant<int, 1>{});
            quarter(std::integral_constant<int, 2>{});
            if constexpr (FIRST) __builtin_amdgcn_s_waitcnt(0x8070);
            else __builtin_amdgcn_s_waitcnt(0x0070);
            ring_barrier();
            __builtin_amdgcn_sched_barrier(0);
            quarter(std::integral_constant<int, 3>{});
            if constexpr (LAST && !(MAGE4_ABL & 2)) {                                       // the next tile's slab 1 (A and W), as the 16 x 16 schedule sends it
                dma_m0(S * G4_STAGE);
                g4_for<8>([&](auto u_) { g4_dma<decltype(u_)::value>(voffA[decltype(u_)::value], a_src); });
                dma_m0(S * G4_STAGE + G4_WOFF);
                g4_for<8>([&](auto u_) { g4_dma<decltype(u_)::value>(voffW[decltype(u_)::value], w_src); });
            }
            if constexpr (!(MAGE4_ABL & 8)) a_src = (j == nk - 3) ? a_next : a_src + 128;
            if constexpr (!(MAGE4_ABL & 8) && LAST) w_src += 128;
        };
#define slab slab32
#endif
        typedef std::integral_constant<int, 0> S0;
        typedef std::integral_constant<int, 1> S1;
        slab(S0{}, std::true_type{}, std::false_type{}, 0);
        slab(S1{}, std::false_type{}, std::false_type{}, 1);
        for (int j = 2; j < nk - 2; j += 2) {
            slab(S0{}, std::false_type{}, std::false_type{}, j);
            slab(S1{}, std::false_type{}, std::false_type{}, j + 1);
        }
        slab(S0{}, std::false_type{}, std::false_type{}, nk - 2);
        slab(S1{}, std::false_type{}, std::true_type{}, nk - 1);
#ifdef MAGE4_MFMA32
#undef slab
#endif
        // the last MFMAs' results must have left the matrix pipe before an accumulator is read (an asm statement gets no hazard padding)
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        G4_STAMP(it, 1);
#if MAGE4_ABL & 1
        {
            f32x4 t = g4_acc_read<0>() + g4_acc_read<17>() + g4_acc_read<63>();
            if (t[0] + t[1] + t[2] + t[3] == 123456.789f) ((float*)g.Y)[0] = t[0] + biasm[0][0][0];
        }
#else
        char* win = smem + G4_RING + wave * 4096;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int n = n0 + 64 + b * 16 + grp * 4;
            biasm[1][b] = *(const f32x4*)(g.bias + n);
            if constexpr (LN == LN_CONSUME) lns[1][b] = *(const f32x4*)(g.ln_colsum + n);
        }
        // gemm.hip's epilogue_lean, 64 x 64 outputs = 64 accumulator registers at a time.  (A software-pipelined rewrite -- 64 rows of staging
        // writes, then their reads under the next 64 rows' arithmetic -- was built and measured 5 % SLOWER per launch: the epilogue is not
        // waiting for its LDS round trips, it is issue-bound on the vector ALU at the clock the K loop's power draw leaves: 1.5 GHz.)
        mage_gemm_desc e = {};
        e.M = 0x7fffffff;
        e.N = LN == LN_PRODUCE ? g.N : 0x7fffffff;
        e.K = g.K;
        e.Y = g.Y;
        e.ldy = g.ldy;
        e.y_dtype = g.y_dtype;
        e.out_h = 1;
        e.out_w = 0x7fffffff;
        e.y_mul_x = 1;
        e.y_off = g.y_off;
        e.y2 = g.y2;
        e.ldy2 = g.ldy2;
        e.ln_part = g.ln_part;
        g4_for<4>([&](auto c_) {
            constexpr int c = decltype(c_)::value, h = c >> 1, mh = c & 1;
            __builtin_amdgcn_sched_barrier(0);
            f32x4 t[4][4];
            g4_for<16>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                t[q >> 2][q & 3] = g4_acc_read<(h * 8 + mh * 4 + (q >> 2)) * 4 + (q & 3), q == 0>();
            });
            [[maybe_unused]] LnConsume lnc;
            if constexpr (LN == LN_CONSUME) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    lnc.mean[a] = ln_mean[mh * 4 + a];
                    lnc.rstd[a] = ln_rstd[mh * 4 + a];
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) lnc.s[b] = lns[h][b];
                if (g.y_dtype == MAGE_F32) epilogue_lean<ACT, float, 4, false, LN>(e, biasm[h], t, m0 + mh * 64, n0 + h * 64, lane, 1, win, 0, &lnc);
                else epilogue_lean<ACT, H16, 4, false, LN>(e, biasm[h], t, m0 + mh * 64, n0 + h * 64, lane, 1, win, 0, &lnc);
            } else if constexpr (LN == LN_DUAL || LN == LN_GELUBWD) {
                epilogue_lean<ACT, unsigned short, 4, false, LN>(e, biasm[h], t, m0 + mh * 64, n0 + h * 64, lane, 1, win, 0);       // bf16 rows (host check)
            } else {
                if (g.y_dtype == MAGE_F32) epilogue_lean<ACT, float, 4, false, LN>(e, biasm[h], t, m0 + mh * 64, n0 + h * 64, lane, 1, win, 0);
                else epilogue_lean<ACT, H16, 4, false, LN>(e, biasm[h], t, m0 + mh * 64, n0 + h * 64, lane, 1, win, 0);
            }
        });
#endif
        G4_STAMP(it, 2);
        // the next tile's first fragments (its slab 0 has been in stage 0 since the last slab's barrier)
        __builtin_amdgcn_sched_barrier(0);
        if (c_tile + nwg8 < chunk1) {
#ifdef MAGE4_MFMA32
#pragma unroll
            for (int b = 0; b < 4; ++b) { rd_w32(0, 0, 0, b); rd_x32(0, 0, 0, b); }
#else
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) rd_w1(0, 0, 0, nt);
#pragma unroll
            for (int m = 0; m < 4; ++m) rd_x1(0, 0, 0, m);
#endif
        }
    }
}

template <int ACT, int EK, int LN, bool RB, bool HF = false>
int launch4(const mage_gemm_desc* d, hipStream_t s, int n_cu) {
    static bool attr[MAGE_MAX_DEVICES] = {false};
    const int dev = mage_device_index();
    MAGE_CHECK_ARG(dev >= 0, "mage_gemm: no current device");
    if (!attr[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm4_kernel<ACT, EK, LN, RB, HF>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS);
        attr[dev] = true;
    }
    Gemm4Args a;
    a.A = d->A;
    a.W = d->W;
    a.Y = d->Y;
    a.bias = d->bias ? d->bias : (const float*)mage_zero_page();      // no bias (the data-gradient GEMMs of training): 16 KiB of zeros, N <= 4096
    a.ln_stats = d->ln_stats;
    a.ln_colsum = d->ln_colsum;
    a.residual = d->residual;
    a.ln_part = d->ln_part;
    a.y2 = d->y2;
    a.M = d->M;
    a.N = d->N;
    a.K = d->K;
    a.lda = d->lda;
    a.ldw = d->ldw ? d->ldw : d->K;
    a.ldy = d->ldy;
    a.ldr = d->ldr;
    a.ldy2 = d->ldy2;
    a.y_dtype = d->y_dtype;
    a.a_off = d->a_off;
    a.y_off = d->y_off;
    a.ntiles_n = d->N / 256;
    a.ntiles = (d->M / 256) * a.ntiles_n;
    // staggered start (gemm.hip, launch_tile): the workgroups of an XCD start in G groups spread over a fraction of one tile period, so that
    // the tiles' output bursts (128-256 KiB per CU at once: HBM-bound when all 256 CUs store together) fall under other groups' K loops
    const int st_groups = mage_options().gemm4_stagger_groups, st_percent = mage_options().gemm4_stagger_percent;
    a.stagger_groups = 0;
    a.stagger_sleeps = 0;
    const int grid = a.ntiles >= n_cu ? n_cu : ((a.ntiles + 7) & ~7);
    if (st_groups > 1 && a.ntiles / grid >= 6) {
        const long period = (long)(d->K / 64) * 3100 + 9000;
        a.stagger_groups = st_groups;
        a.stagger_sleeps = (int)(period * st_percent / 100 / st_groups / 1024);
    }
    hipLaunchKernelGGL((gemm4_kernel<ACT, EK, LN, RB, HF>), dim3(grid), dim3(256), G4_LDS, s, a);
    MAGE_CHECK_LAUNCH("mage_gemm");
    return 1;
}

}  // namespace

namespace {

// ======================================================================================================================================
// gemm_tn4_kernel: the weight-gradient GEMM  P[s][n][k] = sum over the tokens t of slice s of dY[t][n] X[t][k]  (gemm_tn.hip, mage_gemm_tn) on the
// one-wave-per-SIMD schedule: 4 waves x 128 (dY columns) x 128 (X columns), the accumulator file behind inline-asm MFMAs, ONE barrier per
// 64-token slab at the 3/4 point, the loads placed between the MFMAs.  LDS image, transposing fragment reads (ds_read_b64_tr_b16), operand
// roles and the order of the MFMAs per accumulator are gemm_tn_kernel's: the same bits in every partial sum (and in the bias-gradient column
// sums, which the first column of waves of the first X tile takes from the dY fragments with v_dot2_f32_bf16 as before).
struct Tn4Args {
    const unsigned short* A;   // dY [T, lda]
    const unsigned short* B;   // X  [T, ldb]
    float* P;                  // [n_split][N][K]
    float* DB;                 // [n_split][N] or null
    long lda, ldb, T, tps;
    int N, K, n_split, ntk;
};
typedef short g4_tr4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ g4_tr4 g4_lds_tr(const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) g4_tr4*)p); }

template <bool DB>
__global__ __launch_bounds__(256) void gemm_tn4_kernel(const Tn4Args g) {
    constexpr int PART = 32768, STAGE = 2 * PART;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    g4_claim_accumulators();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x / g.n_split, s = blockIdx.x - tile * g.n_split;
    const int tm = tile / g.ntk, tk = tile - tm * g.ntk;
    const long t_begin = (long)s * g.tps;
    const long t_end = t_begin + g.tps < g.T ? t_begin + g.tps : g.T;
    const int nslab = (int)((t_end - t_begin) >> 6);                 // whole slabs only (host: T % 64 == 0), >= 2 (host)
    // ---- loader: wave w moves the 8 token blocks of column strip w (64 columns) of each operand per slab: units w*8 + u
    const int rr = lane >> 3, pc = lane & 7;
    unsigned voffA[8], voffB[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int col_in_strip = (((pc >> 1) ^ (rr & 3) ^ (u & 1)) << 4) + ((pc & 1) << 3);
        voffA[u] = (unsigned)(((long)(u * 8 + rr) * g.lda + wave * 64 + col_in_strip) * 2 + 8192 - u * 1024);
        voffB[u] = (unsigned)(((long)(u * 8 + rr) * g.ldb + wave * 64 + col_in_strip) * 2 + 8192 - u * 1024);
    }
    const char* a_src = (const char*)(g.A + t_begin * g.lda + (long)tm * 256) - 4096;
    const char* b_src = (const char*)(g.B + t_begin * g.ldb + (long)tk * 256) - 4096;
    const long da = 64 * g.lda * 2, db_ = 64 * g.ldb * 2;
    int a_slab = 0, b_slab = 0;                                       // slab index the cursors point at (they stop at the last slab)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 8192 + 4096;
    auto dma_m0 = [&](unsigned lds_group) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds0 + lds_group) : "memory");
    };
    auto a_advance = [&]() __attribute__((always_inline)) {
        const bool more = a_slab + 1 < nslab;
        a_src = more ? a_src + da : a_src;
        a_slab = more ? a_slab + 1 : a_slab;
    };
    auto b_advance = [&]() __attribute__((always_inline)) {
        const bool more = b_slab + 1 < nslab;
        b_src = more ? b_src + db_ : b_src;
        b_slab = more ? b_slab + 1 : b_slab;
    };
    // ---- compute state
    const int wm = wave >> 1, wn = wave & 1;
    const int i15 = lane & 15, grp = lane >> 4;
    const int lane_off = grp * 1024 + (i15 >> 2) * 128 + (i15 & 3) * 8;
    const int sw = (i15 >> 2) ^ (grp & 1);
    g4_tr4 bfh[2][8][2], afh[2][4][2];                 // fragment halves: [buffer][16-column block][tokens 0-3 | 4-7 of the lane group's 8]
    auto rd_b = [&](int buf, int stage, int t, int nt, int h) __attribute__((always_inline)) {       // X columns wn*128 + nt*16
        bfh[buf][nt][h] = g4_lds_tr(smem + stage * STAGE + PART + ((wn * 2 + (nt >> 2)) * 8 + t * 4) * 1024 + lane_off + (((nt & 3) ^ sw) << 5) + h * 512);
    };
    auto rd_a = [&](int buf, int stage, int t, int mt, int h) __attribute__((always_inline)) {       // dY columns wm*128 + mt*16
        afh[buf][mt & 3][h] = g4_lds_tr(smem + stage * STAGE + ((wm * 2 + (mt >> 2)) * 8 + t * 4) * 1024 + lane_off + (((mt & 3) ^ sw) << 5) + h * 512);
    };
    auto frag = [](const g4_tr4 (&h)[2]) __attribute__((always_inline)) {
        return __builtin_bit_cast(u32x4, __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7));
    };
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
    const bf16x2v ones2 = {(__bf16)1.0f, (__bf16)1.0f};
    [[maybe_unused]] float dbs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] auto db_acc = [&](int mt, const g4_tr4 (&h)[2]) __attribute__((always_inline)) {       // column sums of dY from its fragments (gemm_tn_kernel's order)
        const uint2 w0 = __builtin_bit_cast(uint2, h[0]), w1 = __builtin_bit_cast(uint2, h[1]);
        float d = dbs[mt];
        d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, w0.x), ones2, d, false);
        d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, w0.y), ones2, d, false);
        d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, w1.x), ones2, d, false);
        d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, w1.y), ones2, d, false);
        dbs[mt] = d;
    };

    // prologue: slabs 0 and 1 complete, slab 0's first fragments in registers
    dma_m0(0);
    g4_for<8>([&](auto u_) { g4_dma<decltype(u_)::value>(voffA[decltype(u_)::value], a_src); });
    dma_m0(PART);
    g4_for<8>([&](auto u_) { g4_dma<decltype(u_)::value>(voffB[decltype(u_)::value], b_src); });
    a_advance();
    b_advance();
    dma_m0(STAGE);
    g4_for<8>([&](auto u_) { g4_dma<decltype(u_)::value>(voffA[decltype(u_)::value], a_src); });
    dma_m0(STAGE + PART);
    g4_for<8>([&](auto u_) { g4_dma<decltype(u_)::value>(voffB[decltype(u_)::value], b_src); });
    a_advance();
    b_advance();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    ring_barrier();
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) { rd_b(0, 0, 0, nt, 0); rd_b(0, 0, 0, nt, 1); }
#pragma unroll
    for (int m = 0; m < 4; ++m) { rd_a(0, 0, 0, m, 0); rd_a(0, 0, 0, m, 1); }

    // one slab (stage S); accumulator block of (mt, nt) = mt*8 + nt.  Quarters as in gemm4_kernel: (t0, mt 0-3) (t0, mt 4-7) (t1, mt 0-3) | barrier |
    // (t1, mt 4-7); 8 / 24 / 8 / 24 transposing 8-byte fragment reads and the 16 LDS-DMA pieces between the MFMAs
    auto slab = [&](auto S_, auto FIRST_) __attribute__((always_inline)) {
        constexpr int S = decltype(S_)::value;
        constexpr bool FIRST = decltype(FIRST_)::value;
        // ---- Q0: (t0, mt 0-3) from bfh[0], afh[0]; reads dY(t0, mt 4-7) -> afh[1]; X pieces of slab j+1... (sent one slab ahead: see below)
        g4_for<32>([&](auto i_) {
            constexpr int i = decltype(i_)::value, m = i >> 3, nt = i & 7;
            g4_mfma<m * 8 + nt, FIRST, i == 0>(frag(bfh[0][nt]), frag(afh[0][m]));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i < 8) {
                rd_a(1, S, 0, 4 + (i >> 1), i & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (DB && nt == 7) {
                db_acc(m, afh[0][m]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!FIRST && i == 8) {
                dma_m0((S ^ 1) * STAGE + PART);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!FIRST && i > 8 && (i & 1) == 1 && i < 25) {
                g4_dma<((i - 9) >> 1)>(voffB[(i - 9) >> 1], b_src);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        if constexpr (!FIRST) b_advance();
        // ---- Q1: (t0, mt 4-7) from bfh[0], afh[1]; reads dY(t1, mt 0-3) -> afh[0], X(t1) -> bfh[1]
        g4_for<32>([&](auto i_) {
            constexpr int i = decltype(i_)::value, m = i >> 3, nt = i & 7;
            g4_mfma<(4 + m) * 8 + nt, FIRST, i == 0>(frag(bfh[0][nt]), frag(afh[1][m]));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i < 24) {
                if constexpr (i < 8) rd_a(0, S, 1, i >> 1, i & 1);
                else rd_b(1, S, 1, (i - 8) >> 1, i & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (DB && nt == 7) {
                db_acc(4 + m, afh[1][m]);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        // ---- Q2: (t1, mt 0-3) from bfh[1], afh[0]; reads dY(t1, mt 4-7) -> afh[1]
        g4_for<32>([&](auto i_) {
            constexpr int i = decltype(i_)::value, m = i >> 3, nt = i & 7;
            g4_mfma<m * 8 + nt, false, i == 0>(frag(bfh[1][nt]), frag(afh[0][m]));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i < 8) {
                rd_a(1, S, 1, 4 + (i >> 1), i & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (DB && nt == 7) {
                db_acc(m, afh[0][m]);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0) lgkmcnt(0)
        ring_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- Q3: (t1, mt 4-7) from bfh[1], afh[1]; next slab's first fragments from the other stage; dY pieces of slab j+2 -> this stage
        g4_for<32>([&](auto i_) {
            constexpr int i = decltype(i_)::value, m = i >> 3, nt = i & 7;
            g4_mfma<(4 + m) * 8 + nt, false, i == 0>(frag(bfh[1][nt]), frag(afh[1][m]));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i < 24) {
                if constexpr (i < 8) rd_a(0, S ^ 1, 0, i >> 1, i & 1);
                else rd_b(0, S ^ 1, 0, (i - 8) >> 1, i & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (DB && nt == 7) {
                db_acc(4 + m, afh[1][m]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (i == 24) {
                dma_m0(S * STAGE);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        g4_for<8>([&](auto u_) { g4_dma<decltype(u_)::value>(voffA[decltype(u_)::value], a_src); });
        __builtin_amdgcn_sched_barrier(0);
        a_advance();
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    slab(S0{}, std::true_type{});
    int j = 1;
    for (; j + 1 < nslab; j += 2) {
        slab(S1{}, std::false_type{});
        slab(S0{}, std::false_type{});
    }
    if (j < nslab) slab(S1{}, std::false_type{});
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    // ---- epilogue: lane (i, grp) of block (mt, nt) holds row (dY column) mt*16 + i, columns (X columns) nt*16 + grp*4 + {0..3}
    float* out = g.P + ((long)s * g.N + (long)tm * 256 + wm * 128) * g.K + (long)tk * 256 + wn * 128;
    g4_for<64>([&](auto q_) {
        constexpr int q = decltype(q_)::value, mt = q >> 3, nt = q & 7;
        *(f32x4*)(out + (long)(mt * 16 + i15) * g.K + nt * 16 + grp * 4) = g4_acc_read<q, (q & 15) == 0>();
    });
    if constexpr (DB) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            float v = dbs[mt];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (grp == 0) g.DB[(long)s * g.N + (long)tm * 256 + wm * 128 + mt * 16 + i15] = v;
        }
    }
}

}  // namespace

int mage_gemm4h_try(const mage_gemm_desc* d, hipStream_t s, int n_cu);      // gemm4h.hip (applies its own tile-count rule)

// 1 = launched, 0 = not eligible (the caller falls through to the 8-wave kernels), < 0 = error.
// Eligible: bf16 plain GEMMs (no gather, rows not regrouped) with the epilogues y = act(acc + b) (b optional), its LayerNorm-consuming form, or
// training's two c_fc forms (pre-activation + activated rows; data gradient times QuickGELU' of the saved rows),
// act = none | QuickGELU, M and N multiples of 256, K a multiple of 128 in [256, 1024], at least four tiles per CU: the decoder's QKV and c_fc
// at full-loop sizes (12 of the 24 GEMM launches of a decoder pass, 100 of 187 ms of a cfg2 call).  Measured against the 8-phase kernel
// (tools/probes/gemm4_probe.hip, same box, interleaved): QKV +11-12 %, c_fc +7-9 %, N = 512 / K = 512 +4 %; K = 2048 with N = 512 is 9 %
// SLOWER (its A panel is shared by two column tiles only: the loader's HBM latency shows) -- hence the K bound.  The x + Linear(.) kinds
// stay on the 8-wave kernels: their tiles are bounded by the residual / output bursts, not by the K loop.
int mage_gemm4_try(const mage_gemm_desc* d, hipStream_t s) {
    const bool hf = d->dtype == MAGE_F16;              // f16 operands: the bias and LayerNorm-consuming forms of the generation path
    if ((d->dtype != MAGE_BF16 && !hf) || d->n_split > 1) return 0;
    if (mage_options().gemm_no_4w) return 0;           // (mage_set_option: tests run the same product on both kernels in one process)
    if (d->M % 256 || d->N % 256 || d->K % 128 || d->K < 256 || d->K > 1024) return 0;
    if (d->taps_h * d->taps_w != 1 || d->stride != 1 || d->dy0 || d->dx0 || d->in_h != d->out_h || d->in_w != d->out_w || d->a_half) return 0;
    if (d->out_h != 1 || d->out_w < d->M || d->y_mul_x != 1) return 0;                     // plain rows in, plain rows out
    if (d->scale || d->rowadd || d->post_relu || d->res_half) return 0;
    if (d->residual || d->ln_part || d->rowadd) return 0;
    if (!d->bias && (d->N > 4096 || d->ln_stats || mage_zero_page() == nullptr)) return 0;
    if ((d->ln_colsum != nullptr) != (d->ln_stats != nullptr)) return 0;
    // training's two c_fc forms: y2 with QuickGELU = pre-activation rows AND activated rows (LN_DUAL); MAGE_ACT_QUICKGELU_GRAD = the data gradient
    // times QuickGELU'(saved pre-activation rows y2) (LN_GELUBWD); both bf16 rows, no LayerNorm fold
    const bool dual = d->y2 && d->act == MAGE_ACT_QUICKGELU && !d->ln_stats;
    const bool gbwd = d->y2 && d->act == MAGE_ACT_QUICKGELU_GRAD && !d->ln_stats;
    if (d->y2 && !dual && !gbwd) return 0;
    // measured in the training step (rocprofv3, same box): the data-gradient form 736 us per launch on the 8-phase kernel, 840 here (its saved
    // rows are requested inside the epilogue: two waves per SIMD hide that round trip, one does not); the two-output form 787 vs 794.  Both
    // stay on the 8-phase kernel unless the option gemm4_train_forms asks for them here (the instantiations are kept: tests compare the bits)
    if ((dual || gbwd) && !mage_options().gemm4_train_forms) return 0;
    if ((dual || gbwd) && (d->y_dtype != MAGE_BF16 || d->ldy2 % 8 || (((uintptr_t)d->y2) & 15))) return 0;
    if (!dual && !gbwd && d->act != MAGE_ACT_NONE && d->act != MAGE_ACT_QUICKGELU) return 0;
    if (d->y_dtype != MAGE_F32 && d->y_dtype != d->dtype) return 0;
    if (hf && (dual || gbwd)) return 0;
    if (d->ldy % 8 || d->lda % 8 || (((uintptr_t)d->bias | (uintptr_t)d->ln_colsum) & 15) || (((uintptr_t)d->ln_stats) & 7)) return 0;
    const int dev = mage_device_index();
    if (dev < 0) return 0;
    static int n_cu_dev[MAGE_MAX_DEVICES] = {0};
    if (!n_cu_dev[dev]) {
        hipDeviceProp_t p;
        n_cu_dev[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount >= 8) ? (p.multiProcessorCount & ~7) : 256;
    }
    const int n_cu = n_cu_dev[dev];
    // at least four tiles per CU: with two (the incremental loop's c_fc at 16 k rows) the prologue's un-overlapped slab pair and the single
    // tile boundary cost more than the K loop gains (47.5 vs 41 us per launch there): those sizes stay on the lockstep kernel
    const long ntiles = (long)(d->M / 256) * (d->N / 256);
    const bool few_tiles = ntiles < 4L * n_cu;          // gemm4_kernel itself stays above it; its split-half form also runs below (see mage_gemm4h_try)
    const int ldw = d->ldw ? d->ldw : d->K;
    if (((long)d->M + d->a_off) * d->lda * 2 + 16384 >= (1L << 32) || (long)d->N * ldw * 2 + 16384 >= (1L << 32)) return 0;     // 32-bit lane offsets
    if (!dual && !gbwd)
        if (const int r = mage_gemm4h_try(d, s, n_cu)) return r;         // K = 512, 16-bit rows out: the split-half form (gemm4h.hip)
    if (few_tiles) return 0;
    if (dual) return launch4<MAGE_ACT_QUICKGELU, EK_BIAS, LN_DUAL, false>(d, s, n_cu);
    if (gbwd) return launch4<MAGE_ACT_NONE, EK_BIAS, LN_GELUBWD, false>(d, s, n_cu);
    if (hf) {
        if (d->ln_stats) {
            if (d->act == MAGE_ACT_NONE) return launch4<MAGE_ACT_NONE, EK_BIAS, LN_CONSUME, false, true>(d, s, n_cu);
            return launch4<MAGE_ACT_QUICKGELU, EK_BIAS, LN_CONSUME, false, true>(d, s, n_cu);
        }
        if (d->act == MAGE_ACT_NONE) return launch4<MAGE_ACT_NONE, EK_BIAS, LN_NONE, false, true>(d, s, n_cu);
        return launch4<MAGE_ACT_QUICKGELU, EK_BIAS, LN_NONE, false, true>(d, s, n_cu);
    }
    if (d->ln_stats) {
        if (d->act == MAGE_ACT_NONE) return launch4<MAGE_ACT_NONE, EK_BIAS, LN_CONSUME, false>(d, s, n_cu);
        return launch4<MAGE_ACT_QUICKGELU, EK_BIAS, LN_CONSUME, false>(d, s, n_cu);
    }
    if (d->act == MAGE_ACT_NONE) return launch4<MAGE_ACT_NONE, EK_BIAS, LN_NONE, false>(d, s, n_cu);
    return launch4<MAGE_ACT_QUICKGELU, EK_BIAS, LN_NONE, false>(d, s, n_cu);
}

// mage_gemm_tn on the one-wave-per-SIMD schedule: 1 = launched, 0 = not eligible (gemm_tn.hip then runs its 8-wave kernel)
int mage_gemm_tn4_try(const void* dY, int64_t lda, const void* X, int64_t ldb, int64_t T, int32_t N, int32_t K, int32_t n_split, int64_t tps,
                      float* partials, float* db_partials, hipStream_t stream) {
    if (mage_options().gemm_no_4w) return 0;
    if (T % 64 || tps % 64 || tps < 128 || (T - (int64_t)(n_split - 1) * tps) < 128) return 0;          // whole slabs, at least two per slice
    if (lda * 2 * 64 + 16384 >= (1L << 31) || ldb * 2 * 64 + 16384 >= (1L << 31)) return 0;              // 32-bit lane offsets inside a slab
    const int dev = mage_device_index();
    if (dev < 0) return 0;
    static bool attr[MAGE_MAX_DEVICES] = {false};
    if (!attr[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm_tn4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_tn4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr[dev] = true;
    }
    Tn4Args a;
    a.A = (const unsigned short*)dY;
    a.B = (const unsigned short*)X;
    a.P = partials;
    a.DB = db_partials;
    a.lda = lda;
    a.ldb = ldb;
    a.T = T;
    a.tps = tps;
    a.N = N;
    a.K = K;
    a.n_split = n_split;
    a.ntk = K / 256;
    const long grid = (long)(N / 256) * (K / 256) * n_split;
    if (db_partials) hipLaunchKernelGGL(gemm_tn4_kernel<true>, dim3((unsigned)grid), dim3(256), 128 * 1024, stream, a);
    else hipLaunchKernelGGL(gemm_tn4_kernel<false>, dim3((unsigned)grid), dim3(256), 128 * 1024, stream, a);
    MAGE_CHECK_LAUNCH("mage_gemm_tn");
    return 1;
}
