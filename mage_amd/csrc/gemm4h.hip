// gemm4h_kernel: gemm4_kernel's 256 x 256 tile with the epilogue UNDER the K loop (round 6; DESIGN.md section 9 item 0 of round 5).
//
// gemm4_kernel (gemm4.hip) gives each of its four waves 128 x 128 outputs = the whole accumulator file, so a tile's epilogue (LayerNorm fold,
// QuickGELU, pack, stores: 36 % of a c_fc tile) runs with the matrix pipe idle -- nothing else is resident on the SIMD.
// Here the wave's block is two HALVES of 64 rows (128 accumulator registers each) that alternate:
//     pass (tile t, half 0): K loop into a[..] of half 0   ||  the finished half 1 of tile t-1 leaves through the epilogue
//     pass (tile t, half 1): K loop into a[..] of half 1   ||  half 0 of tile t leaves
// One pass = 8 steps (K = 512: 8 slabs of 64) of 64 MFMAs; one epilogue PIECE (16 rows x 64 columns = 4 accumulator blocks) per step, cut into
// single instructions ("ops", asm statements: hipcc keeps those in source order, plain arithmetic it sinks to the consumer) that are placed
// BETWEEN the step's MFMAs by issue cost (profiles/r06_mfma_valu_coissue.txt: beside a 16-cycle MFMA one wave issues two plain VALU
// instructions or one of exp / rcp / cvt_pk / accvgpr_read for free).  The arithmetic is epilogue_lean's, operation for operation
// (fma(-mean, s, acc), fma(rstd, ., c), the QuickGELU chain, v_cvt_pk): the same bits as every other GEMM kernel of the library.
//
// No LDS transposition: the MFMA operand roles are swapped against gemm4_kernel (A-operand = activation rows; same products, same k order,
// same bits) and the W rows are gathered into a permuted LDS order, so that a lane holds 4 CONSECUTIVE output columns of 4 rows and the 16
// lanes of a group one whole 128-byte row: rows are stored straight from registers (global_store_dwordx2, scalar base + one per-lane offset).
// (A ds_write between the fragment reads cost 100 cycles in this loop: LDS operations of a wave execute in order.)
//
// Cost of the split: a 64 x 128 half reads 12 operand fragments per 32 MFMAs (gemm4: 16 per 64) and the W slab is fetched once per half
// (1.5x the LDS-DMA bytes per MFMA, from L2): the K loop alone runs 20.3 cycles per MFMA against gemm4's 17.7.  What it buys (measured,
// profiles/r06_gemm4h_split_half.txt): c_fc (LayerNorm fold + QuickGELU) +9 %; QKV (LayerNorm fold only) +0..5 %.  LDS: A ring 4 x 16 KB (the
// pass's 128 A rows x 64 k), W ring 2 x 32 KB, 2 x 2 KB per wave of epilogue constants (the tile's (mean, rstd) rows, bias and LayerNorm
// column sums, brought in by LDS-DMA one pass ahead, read one piece ahead of their use): 144 KB.
//
// Step q (stage indices are compile time: a pass has 8 steps, the rings 4 and 2 stages):
//     t0: 32 MFMAs from fragment buffer 0;  reads: fragments (q, t1) -> buffer 1
//     s_waitcnt vmcnt(N) lgkmcnt(0);  s_barrier          -- W(q+1) and A(q+1) of every wave have landed, stages of step q are free
//     t1: 32 MFMAs from buffer 1;  reads: fragments (q+1, t0) -> buffer 0 (other stages);  LDS-DMA: W(q+2) x 8, then A(q+4) x 4
// Memory operations retire in order, so the wait is COUNTED: N = the operations issued after the W pieces it needs (the four A pieces, the
// row stores behind them, the constants' pieces) -- they stay in flight across the barrier; the A pieces (HBM) are forced home one barrier
// later, two steps after their issue.
#include "gemm_shared.h"
#include "gemm4_regs.h"

namespace {

struct Gemm4hArgs {
    const void* A;
    const void* W;
    void* Y;
    const float* bias;
    const float* ln_stats;
    const float* ln_colsum;
    int M, N, lda, ldw, ldy, a_off, y_off;
    int ntiles_n, ntiles;
};

#ifndef MAGE4H_ABL
#define MAGE4H_ABL 0      // tuning builds (tools/probes/gemm4h_probe.hip), bit flags: 1 = no epilogue (K loop only), 2 = epilogue ops but no global stores
#endif
#ifndef MAGE4H_OPMASK
#define MAGE4H_OPMASK 7   // tuning builds, which op classes of a piece are emitted: 1 = accumulator reads, 2 = arithmetic, 4 = the constants' LDS reads
                          // (timing only: the outputs are garbage unless all are on)
#endif
#ifndef MAGE4H_R0
#define MAGE4H_R0 2       // first slot of the fragment reads in each half step
#endif

#ifdef MAGE4H_STAMP
// tuning build: per workgroup (wave 0) shader-clock and 100 MHz wall-clock stamps around the pass loop: the clock the chip holds while this kernel runs
__device__ unsigned long long h_stamps[256 * 4];
#endif

constexpr int H_ASTG = 16384, H_WSTG = 32768, H_WOFF = 4 * H_ASTG, H_RING = H_WOFF + 2 * H_WSTG;      // 128 KB of operand rings
constexpr int H_CONST_OFF = H_RING;                     // 4 waves x 2 sets x 2 KB: [stats 128 rows x 8 B | bias 128 x 4 B | colsum 128 x 4 B]
constexpr int H_LDS = H_CONST_OFF + 4 * 4096;           // 144 KB (no row staging: the rows leave the accumulator layout directly, see below)
constexpr int H_CSTEP = 2;                              // the step of a tile's half-0 pass that requests the tile's epilogue constants

template <int R, bool CLOB>
__device__ __forceinline__ float h_acc1() {
    float v;
    if constexpr (CLOB) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(R) : G4_ALL_ACC);
    else asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(R));
    return v;
}
typedef unsigned h_u32x2 __attribute__((ext_vector_type(2)));
// streaming store of 8 bytes per lane at (scalar base + per-lane 32-bit offset)
__device__ __forceinline__ void h_store2(unsigned voff, h_u32x2 d, const char* base) {
    asm volatile("global_store_dwordx2 %0, %1, %2 nt" ::"v"(voff), "v"(d), "s"(base) : "memory");
}
__device__ __forceinline__ void h_sink(unsigned a, unsigned b, const void* p) { asm volatile("" ::"v"(a), "v"(b), "v"(p)); }
// s_waitcnt immediate of gfx9: vmcnt(vm) lgkmcnt(0), expcnt untouched
constexpr int h_wait_imm(int vm) { return (((vm >> 4) & 3) << 14) | (vm & 15) | 0x70; }


// Placement of a piece's ops in the 64 MFMA slots of its step, by ISSUE COST (tools/probes/mfma_valu_probe.hip, profiles/r06_mfma_valu_coissue.txt:
// beside back-to-back 16-cycle MFMAs one wave issues ~8 more cycles per MFMA for free -- two plain VALU instructions (4 cycles each) or ONE of
// v_exp / v_rcp / v_cvt_pk / v_accvgpr_read (8 each); past that every instruction delays the next MFMA by its full issue time).
// Unit = 4 cycles.  A slot holds 2 units; 1 if it already carries a fragment read, an LDS-DMA piece or a store.
struct HPlan { int lo[65]; int st[4]; };           // op range per slot; the slot of each of the piece's four row stores
// A piece = 4 GROUPS (one per accumulator register e = the row 4g + e of the lane group's four) of OPG ops over the piece's 4 blocks j:
//   0-3 accumulator reads, 4 the NEXT piece's constants (group 3 only), LN ops, activation, pack (2 converts) + the group's row store
template <bool CONS, bool GELU>
constexpr int h_op_cost(int K) {
    constexpr int LNOPS = CONS ? 8 : 4, ACTOPS = GELU ? 20 : 0, LNB = 5, GB = LNB + LNOPS, PK = GB + ACTOPS, OPG = PK + 2;
    const int e = K / OPG, o = K % OPG;
    if (o < 4) return 2;
    if (o == 4) return e == 3 ? (CONS ? 4 : 1) : 0;
    if (o < GB) return 1;
    if (o < PK) { const int a = (o - GB) >> 2; return (a == 1 || a == 3) ? 2 : 1; }
    return o == PK ? 2 : 4;
}
template <bool CONS, bool GELU>
constexpr HPlan h_make_plan(int S, int R0, bool consts_step) {
    constexpr int LNOPS = CONS ? 8 : 4, ACTOPS = GELU ? 20 : 0, OPG = 5 + LNOPS + ACTOPS + 2, NOPS = 4 * OPG;
    HPlan p = {};
    int k = 0, rem = 0;
    for (int i = 0; i < NOPS; ++i) rem += h_op_cost<CONS, GELU>(i);
    for (int sl = 0; sl < 64; ++sl) {
        p.lo[sl] = k;
        int cap = 2;
        if (sl < 32) {
            if (sl >= R0 && sl < R0 + 12) cap = 1;
        } else {
            const int i = sl - 32;
            if (i < 24) cap = 1;
            if (i == 24 && consts_step) cap = 0;
        }
        const int slots_left = 64 - sl;
        const int need = (rem + slots_left - 1) / slots_left;          // what this slot must take for the rest to fit evenly
        const int target = need > cap ? need : cap;
        int used = 0;
        while (k < NOPS) {
            const int c = h_op_cost<CONS, GELU>(k);
            if (used > 0 && used + c > target) break;
            if (used == 0 && cap == 0 && need <= 2) break;
            if (k % OPG == OPG - 1) p.st[k / OPG] = sl;
            used += c;
            rem -= c;
            ++k;
        }
    }
    p.lo[64] = k == NOPS ? NOPS : -1;
    return p;
}
// stores of a step that are younger than its last W piece (issued at or after slot 46) / issued before its barrier (slots 0..31)
constexpr int h_st_after_w(const HPlan& p) { int n = 0; for (int e = 0; e < 4; ++e) n += p.st[e] >= 46; return n; }
constexpr int h_st_t0(const HPlan& p) { int n = 0; for (int e = 0; e < 4; ++e) n += p.st[e] < 32; return n; }

template <int ACT, int LN, bool HF>
__global__ __launch_bounds__(256) void gemm4h_kernel(const Gemm4hArgs g) {
    static_assert(LN == LN_NONE || LN == LN_CONSUME, "the generation path's epilogues");
    static_assert(ACT == MAGE_ACT_NONE || ACT == MAGE_ACT_QUICKGELU, "act none | QuickGELU");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    g4_claim_accumulators();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // persistent tile schedule of gemm4_kernel: workgroup b runs on XCD b % 8; each XCD owns a contiguous chunk of the tile list
    const int nwg8 = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int q8 = g.ntiles >> 3, r8 = g.ntiles & 7;
    const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int chunk1 = chunk0 + q8 + (xcd < r8 ? 1 : 0);
    const int tile0 = chunk0 + li;
    if (tile0 >= chunk1) return;
    const int n_my = (chunk1 - tile0 + nwg8 - 1) / nwg8;

    // ---- loader.  A unit = 8 rows x 128 B = one wave-wide LDS-DMA.  Per step this wave moves 4 A units (rows wave*32.. of the pass's 128-row
    // image: row = wm*64 + row in the half) and 8 W units (rows wave*64..), gemm4's addressing: M0 once per group, the instruction's immediate
    // (u - 4) KiB selects the unit on both sides.
    const int lr = lane >> 3, lp = lane & 7;
    unsigned voffA[4], voffW[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = u * 8 + lr;
        voffA[u] = (unsigned)(r * g.lda * 2 + ((lp ^ ((r >> 1) & 7)) << 4) + 8192 - u * 1024);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        // LDS row slot r of the wave's 64-row strip (fragment j = r >> 4, lane row r & 15) holds W row 4 (r & 15) + j of the strip: with the MFMA
        // operand roles of this kernel (below) lane l15 then owns the 4 CONSECUTIVE output columns 4 l15 + j of a 64-column piece
        const int r = u * 8 + lr, rs = 4 * (r & 15) + (r >> 4);
        voffW[u] = (unsigned)(rs * g.ldw * 2 + ((lp ^ ((r >> 1) & 7)) << 4) + 8192 - u * 1024);
    }
    auto a_base = [&](int tile, int half) {
        return (const char*)g.A + ((long)((tile / g.ntiles_n) * 256 + (wave >> 1) * 128 + half * 64 + (wave & 1) * 32) + g.a_off) * g.lda * 2 - 4096;
    };
    auto w_base = [&](int tile) { return (const char*)g.W + (long)((tile % g.ntiles_n) * 256 + wave * 64) * g.ldw * 2 - 4096; };
    const unsigned smem_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lds_a0 = smem_addr + wave * 4096 + 4096;
    const unsigned lds_w0 = smem_addr + H_WOFF + wave * 8192 + 4096;
    auto dma_m0 = [&](unsigned addr) __attribute__((always_inline)) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(addr) : "memory"); };
    const char* a_ptr = a_base(tile0, 0);
    const char* w_ptr = w_base(tile0);
    const char* a_next = a_ptr;
    const char* w_next = w_ptr;

    // ---- compute state: wave (wm, wn) owns rows [wm*128, +128) x columns [wn*128, +128) of the tile; half h = rows [wm*128 + 64h, +64)
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, grp = lane >> 4;
    const int rsw = (l15 >> 1) & 7;
    const int pc[2] = {((grp + 0) ^ rsw) * 16, ((grp + 4) ^ rsw) * 16};
    const int x_off = (wm * 64 + l15) * 128;
    const int w_off = H_WOFF + (wn * 128 + l15) * 128;
    u32x4 wf[2][8], xf[2][4];
    auto rd_w = [&](int buf, int stage, int t, int nt) __attribute__((always_inline)) {
        wf[buf][nt] = *(const u32x4*)(smem + stage * H_WSTG + w_off + nt * 2048 + pc[t]);
    };
    auto rd_x = [&](int buf, int stage, int t, int mt) __attribute__((always_inline)) {
        xf[buf][mt] = *(const u32x4*)(smem + stage * H_ASTG + x_off + mt * 2048 + pc[t]);
    };

    // ---- epilogue state (one piece in flight): the drained half's coordinates, the piece's values and store pointer
    constexpr bool CONS = LN == LN_CONSUME;
    constexpr bool GELU = ACT == MAGE_ACT_QUICKGELU;
    constexpr int LNOPS = CONS ? 8 : 4, ACTOPS = GELU ? 20 : 0;
    constexpr int LNB = 5, GB = LNB + LNOPS, PK = GB + ACTOPS, OPG = PK + 2;
    constexpr int NOPS = 4 * OPG;
    constexpr int NC = CONS ? 2 : 1;                  // LDS-DMA pieces of a tile's epilogue constants
    // MFMA operand roles: A-operand = the activation fragment, B-operand = the W fragment (gemm4_kernel: the other way round; the products and
    // their k order are the same, so are the bits).  The accumulator block (16-row tile mt, fragment nt = 4 ch + j) then holds, in lane (l15, g)
    // and register e, the output of ROW mt*16 + 4g + e and COLUMN ch*64 + 4 l15 + j: a lane's four blocks j are 4 consecutive columns, the 16
    // lanes of a group one whole 128-byte row -- the rows are stored straight from the registers (global_store_dwordx2, 4 rows x 128 B per
    // instruction), no LDS transposition (a ds_write beside the fragment reads cost 100 cycles: profiles/r06_gemm4h_split_half.txt).
    [[maybe_unused]] float ea[4], eu[4];
    [[maybe_unused]] f32x4 esb[2], ebb[2], em[2][2];   // by piece parity: column sums / bias of the lane's 4 columns, (mean, rstd) of its 4 rows
    [[maybe_unused]] unsigned epk0 = 0;
    // row stores: global_store_dwordx2 voff, data, s[base]: ONE per-lane byte offset for the whole kernel (row 4g, column 4 l15 of a 16 x 64 piece),
    // the piece's / row's position in the SCALAR base (s_add: free beside the MFMAs; 64-bit vector address arithmetic per store is not)
    const unsigned evoff = (unsigned)(((long)4 * grp * g.ldy + 4 * l15) * 2);
    [[maybe_unused]] const char* eyp = (const char*)g.Y;         // scalar: the piece's block
    [[maybe_unused]] const char* eybase = (const char*)g.Y;      // scalar: the drained half's 64 x 128 block of this wave
    [[maybe_unused]] const char* ecb = smem + H_CONST_OFF + wave * 4096;      // the drained tile's constants
    [[maybe_unused]] const char* ecbn = ecb;                                  // ... of the NEXT pass's drained tile
    const long ldyb = (long)g.ldy * 2;                 // bytes per output row
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

    // the constants of piece P of half DH from set cb, read ONE PIECE AHEAD of their use (the wave would sit in s_waitcnt lgkmcnt otherwise)
    auto epi_consts = [&](auto DH_, auto P_, const char* cb) __attribute__((always_inline)) {
        constexpr int DH = decltype(DH_)::value, P = decltype(P_)::value, mt = P & 3, ch = P >> 2;
        if constexpr (!(MAGE4H_OPMASK & 4)) return;
        const char* cp = cb + 1024 + (ch * 64 + 4 * l15) * 4;
        ebb[P & 1] = *(const f32x4*)cp;
        if constexpr (CONS) {
            esb[P & 1] = *(const f32x4*)(cp + 512);
            const char* mp = cb + (DH * 64 + mt * 16 + 4 * grp) * 8;
            em[P & 1][0] = *(const f32x4*)mp;
            em[P & 1][1] = *(const f32x4*)(mp + 16);
        }
    };
    auto epi_ptr = [&](auto P_) __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value;
        eyp = eybase + (P & 3) * 16 * ldyb + (P >> 2) * 128;
    };
    // op K of piece P of the drained half DH (its 16-row tile mt = P & 3, 64-column half ch = P >> 2): group e = K / OPG
    auto epi_op = [&](auto DH_, auto P_, auto K_) __attribute__((always_inline)) {
        constexpr int DH = decltype(DH_)::value, P = decltype(P_)::value, K = decltype(K_)::value;
        constexpr int mt = P & 3, ch = P >> 2, e = K / OPG, o = K % OPG;
        // every arithmetic op is ONE instruction as an asm statement: hipcc keeps asm volatile statements in source order, whereas plain
        // arithmetic written between the MFMAs sinks to its consumer (measured: the whole block's chain landed in one MFMA slot).
        // Hazards are this code's business: a transcendental's result is read >= 3 instructions later (gfx950 needs one wait state).
        if constexpr (o < 4) {
            constexpr int blk = (ch * 8 + DH * 4 + mt) * 4 + o;
            if constexpr (MAGE4H_OPMASK & 1) ea[o] = h_acc1<4 * blk + e, o == 0>();
        } else if constexpr (o == 4) {
            if constexpr (e == 3) {
                if constexpr (P < 7) epi_consts(DH_, std::integral_constant<int, P + 1>{}, ecb);
                else epi_consts(std::integral_constant<int, 1 - DH>{}, std::integral_constant<int, 0>{}, ecbn);
            }
        } else if constexpr (!(MAGE4H_OPMASK & 2) && o < PK + 1) {
        } else if constexpr (o < LNB + 4) {
            constexpr int j = o - LNB;
            if constexpr (CONS) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(ea[j]) : "v"(em[P & 1][e >> 1][(e & 1) * 2]), "v"(esb[P & 1][j]));
            else asm volatile("v_add_f32 %0, %0, %1" : "+v"(ea[j]) : "v"(ebb[P & 1][j]));
        } else if constexpr (CONS && o < LNB + 8) {
            constexpr int j = o - LNB - 4;
            asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(ea[j]) : "v"(em[P & 1][e >> 1][(e & 1) * 2 + 1]), "v"(ebb[P & 1][j]));
        } else if constexpr (o < PK) {
            constexpr int a = (o - GB) >> 2, j = (o - GB) & 3;
            if constexpr (a == 0) asm volatile("v_mul_f32 %0, 0xc01d265f, %1" : "=v"(eu[j]) : "v"(ea[j]));        // -1.702 log2(e) x
            else if constexpr (a == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(eu[j]));
            else if constexpr (a == 2) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(eu[j]));
            else if constexpr (a == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(eu[j]));
            else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(ea[j]) : "v"(eu[j]));
        } else if constexpr (o == PK) {
            if constexpr (HF) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(epk0) : "v"(ea[0]), "v"(ea[1]));
            else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(epk0) : "v"(ea[0]), "v"(ea[1]));
        } else {
            unsigned pk1 = 0;
            if constexpr (!(MAGE4H_OPMASK & 2)) {
            } else if constexpr (HF) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk1) : "v"(ea[2]), "v"(ea[3]));
            else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk1) : "v"(ea[2]), "v"(ea[3]));
#if !(MAGE4H_ABL & 2)
            h_store2(evoff, u32x2_t{epk0, pk1}, eyp + e * ldyb);                                       // row mt*16 + 4g + e: 16 lanes x 8 B = its 128 bytes
#else
            h_sink(epk0, pk1, eyp);
#endif
        }
    };

    // ---- prologue: W(0), W(1), A(0..3) of the first pass; everything home; slab 0's first fragments in registers
    dma_m0(lds_w0);
    g4_for<8>([&](auto u_) { g4_dma<decltype(u_)::value>(voffW[decltype(u_)::value], w_ptr); });
    dma_m0(lds_w0 + H_WSTG);
    g4_for<8>([&](auto u_) { g4_dma<decltype(u_)::value>(voffW[decltype(u_)::value], w_ptr + 128); });
    w_ptr += 256;
    g4_for<4>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        dma_m0(lds_a0 + s * H_ASTG);
        g4_for<4>([&](auto u_) { g4_dma<decltype(u_)::value>(voffA[decltype(u_)::value], a_ptr + s * 128); });
    });
    a_ptr += 512;
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
    ring_barrier();
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) rd_w(0, 0, 0, nt);
#pragma unroll
    for (int m = 0; m < 4; ++m) rd_x(0, 0, 0, m);

    // One step.  H = the accumulating half, S = the slab (0..7), EPI = the other half leaves meanwhile, CONSTS = this pass requests the epilogue
    // constants of its tile (half-0 passes), prev_stores = the previous pass was an epilogue pass, i.e. its last step left row stores behind its
    // last W piece (wave-uniform, read by S == 0 only), cpar = parity of this pass's tile (selects the constants' set), tm / tn = this pass's tile.
    auto step = [&](auto H_, auto S_, auto EPI_, auto CONSTS_, bool prev_stores, int cpar, int tm, int tn) __attribute__((always_inline)) {
        constexpr int H = decltype(H_)::value, S = decltype(S_)::value;
        constexpr bool EPI = decltype(EPI_)::value && !(MAGE4H_ABL & 1), CONSTS = decltype(CONSTS_)::value && !(MAGE4H_ABL & 1);
        constexpr int SA = S & 3, SW = S & 1, SA1 = (S + 1) & 3, SW1 = (S + 1) & 1;
        constexpr int R0 = MAGE4H_R0;
        [[maybe_unused]] constexpr HPlan plan = h_make_plan<CONS, GELU>(S, R0, CONSTS && S == H_CSTEP);
        static_assert(plan.lo[64] == NOPS, "every op of the piece has a slot");
        auto ops_at = [&](auto SL_) __attribute__((always_inline)) {
            constexpr int sl = decltype(SL_)::value;
            if constexpr (!EPI && !(MAGE4H_ABL & 1) && S == 7 && sl == 44) {        // the workgroup's first pass: what the first piece of the next one finds prefetched
                epi_consts(std::integral_constant<int, H>{}, std::integral_constant<int, 0>{}, ecbn);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (EPI) {
                if constexpr (sl == 0) {
                    epi_ptr(S_);
                    __builtin_amdgcn_sched_barrier(0);
                }
                constexpr int lo = plan.lo[sl], hi = plan.lo[sl + 1];
                g4_for<hi - lo>([&](auto k_) {
                    epi_op(std::integral_constant<int, 1 - H>{}, S_, std::integral_constant<int, lo + decltype(k_)::value>{});
                });
                if constexpr (hi > lo) __builtin_amdgcn_sched_barrier(0);
            }
        };
        // ---- t0
        g4_for<32>([&](auto i_) {
            constexpr int i = decltype(i_)::value, m = i >> 3, nt = i & 7;
            g4_mfma<((nt >> 2) * 8 + 4 * H + m) * 4 + (nt & 3), S == 0, i == 0, HF>(xf[0][m], wf[0][nt]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i >= R0 && i < R0 + 12) {
                if constexpr (i - R0 < 4) rd_x(1, SA, 1, i - R0);
                else rd_w(1, SW, 1, i - R0 - 4);
                __builtin_amdgcn_sched_barrier(0);
            }
            ops_at(i_);
        });
        // W(q+1) (and everything older: A(q+1), the stores before it) home; in flight behind it: A(q+3)'s 4 pieces, the constants' pieces if the
        // previous step sent them, the previous step's stores issued after its last W piece, this step's stores so far
        {
            constexpr bool prev_consts = S >= 1 && (S - 1) == H_CSTEP && CONSTS;
            constexpr HPlan prev = h_make_plan<CONS, GELU>(S >= 1 ? S - 1 : 7, R0, prev_consts);
            constexpr int n_base = 4 + (prev_consts ? NC : 0) + (EPI ? h_st_t0(plan) : 0);
            constexpr int n_prev = h_st_after_w(prev);
            if constexpr (S == 0) {
                if (prev_stores) __builtin_amdgcn_s_waitcnt(h_wait_imm(n_base + n_prev));
                else __builtin_amdgcn_s_waitcnt(h_wait_imm(n_base));
            } else {
                __builtin_amdgcn_s_waitcnt(h_wait_imm(n_base + (EPI ? n_prev : 0)));
            }
        }
        ring_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- t1
        g4_for<32>([&](auto i_) {
            constexpr int i = decltype(i_)::value, m = i >> 3, nt = i & 7;
            g4_mfma<((nt >> 2) * 8 + 4 * H + m) * 4 + (nt & 3), false, i == 0, HF>(xf[1][m], wf[1][nt]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((i & 1) == 1 && i < 24) {                 // fragments (q+1, t0): the next step's stages
                constexpr int k = i >> 1;
                if constexpr (k < 4) rd_x(0, SA1, 0, k);
                else rd_w(0, SW1, 0, k - 4);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (i == 0) dma_m0(lds_w0 + SW * H_WSTG);
            if constexpr ((i & 1) == 0 && i < 16) g4_dma<(i >> 1)>(voffW[i >> 1], w_ptr);            // W(q+2): 8 pieces
            if constexpr (i == 16) dma_m0(lds_a0 + SA * H_ASTG);
            if constexpr ((i & 1) == 0 && i >= 16 && i < 24) g4_dma<((i - 16) >> 1)>(voffA[(i - 16) >> 1], a_ptr);      // A(q+4): 4 pieces
            if constexpr ((i & 1) == 0 && i < 24) __builtin_amdgcn_sched_barrier(0);
            if constexpr (CONSTS && S == H_CSTEP) {                 // this tile's epilogue constants -> set cpar (first read one pass later)
                if constexpr (i == 24) {
                    const unsigned cset = smem_addr + H_CONST_OFF + wave * 4096 + cpar * 2048;
                    if constexpr (CONS) {
                        dma_m0(cset);
                        const char* sp = (const char*)(g.ln_stats + 2 * (long)(tm * 256 + wm * 128));
                        asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"((unsigned)(lane * 16)), "s"(sp) : "memory");
                    }
                    dma_m0(cset + 1024);
                    const float* vp = ((CONS && lane >= 32) ? g.ln_colsum : g.bias) + tn * 256 + wn * 128 + (lane & 31) * 4;
                    asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(vp) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            ops_at(std::integral_constant<int, 32 + i>{});
        });
        // cursors: W runs 2 steps ahead (slab (S+3) & 7 next), A 4 (slab (S+5) & 7 next); at a pass's end they jump to the next pass's rows
        w_ptr = (S == 5) ? w_next : w_ptr + 128;
        a_ptr = (S == 3) ? a_next : a_ptr + 128;
    };
    auto pass = [&](auto H_, auto EPI_, bool prev_stores, int cpar, int tm, int tn) __attribute__((always_inline)) {
        constexpr bool C = decltype(H_)::value == 0;
        typedef std::integral_constant<bool, C> CT;
        step(H_, std::integral_constant<int, 0>{}, EPI_, CT{}, prev_stores, cpar, tm, tn);
        step(H_, std::integral_constant<int, 1>{}, EPI_, CT{}, false, cpar, tm, tn);
        step(H_, std::integral_constant<int, 2>{}, EPI_, CT{}, false, cpar, tm, tn);
        step(H_, std::integral_constant<int, 3>{}, EPI_, CT{}, false, cpar, tm, tn);
        step(H_, std::integral_constant<int, 4>{}, EPI_, CT{}, false, cpar, tm, tn);
        step(H_, std::integral_constant<int, 5>{}, EPI_, CT{}, false, cpar, tm, tn);
        step(H_, std::integral_constant<int, 6>{}, EPI_, CT{}, false, cpar, tm, tn);
        step(H_, std::integral_constant<int, 7>{}, EPI_, CT{}, false, cpar, tm, tn);
    };
    // the drained half (tile (dtm, dtn), half dh) of the pass that follows: its constants' set and its output block
    auto drain_of = [&](int dtm, int dtn, int dh, int dpar, int npar) __attribute__((always_inline)) {
        ecb = smem + H_CONST_OFF + wave * 4096 + dpar * 2048;
        ecbn = smem + H_CONST_OFF + wave * 4096 + npar * 2048;
        eybase = (const char*)g.Y + (((long)(dtm * 256 + wm * 128 + dh * 64) + g.y_off) * g.ldy + dtn * 256 + wn * 128) * 2;
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
#ifdef MAGE4H_STAMP
    if (tid == 0 && blockIdx.x < 256) {
        h_stamps[blockIdx.x * 4 + 0] = __builtin_readcyclecounter();
        h_stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    int tile = tile0, ptm = 0, ptn = 0;
    for (int ts = 0; ts < n_my; ++ts, tile += nwg8) {
        const int tm = tile / g.ntiles_n, tn = tile - tm * g.ntiles_n;
        MAGE_DASSERT(tile >= 0 && tile < g.ntiles);
        // ---- half 0 of this tile; half 1 of the previous tile leaves
        a_next = a_base(tile, 1);
        w_next = w_base(tile);
        if (ts == 0) {
            ecbn = smem + H_CONST_OFF + wave * 4096;                   // the next pass drains this tile's half 0: set 0
            pass(I0{}, std::false_type{}, false, ts & 1, tm, tn);
        } else {
            drain_of(ptm, ptn, 1, (ts - 1) & 1, ts & 1);
            pass(I0{}, std::true_type{}, true, ts & 1, tm, tn);
        }
        // ---- half 1; half 0 leaves
        {
            const int ntile = ts + 1 < n_my ? tile + nwg8 : tile;      // past the end: re-fetch this tile (never read)
            a_next = a_base(ntile, 0);
            w_next = w_base(ntile);
        }
        drain_of(tm, tn, 0, ts & 1, ts & 1);
        pass(I1{}, std::true_type{}, ts > 0, ts & 1, tm, tn);
        ptm = tm;
        ptn = tn;
    }
#ifdef MAGE4H_STAMP
    if (tid == 0 && blockIdx.x < 256) {
        h_stamps[blockIdx.x * 4 + 2] = __builtin_readcyclecounter();
        h_stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#if !(MAGE4H_ABL & 1)
    // ---- the last half: its eight pieces, nothing to hide them under
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    drain_of(ptm, ptn, 1, (n_my - 1) & 1, (n_my - 1) & 1);
    g4_for<8>([&](auto p_) {
        epi_ptr(p_);
        g4_for<NOPS>([&](auto k_) { epi_op(I1{}, p_, k_); });
    });
#endif
}

template <int ACT, int LN, bool HF>
int launch4h(const mage_gemm_desc* d, hipStream_t s, int n_cu) {
    static bool attr[MAGE_MAX_DEVICES] = {false};
    const int dev = mage_device_index();
    MAGE_CHECK_ARG(dev >= 0, "mage_gemm: no current device");
    if (!attr[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm4h_kernel<ACT, LN, HF>, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS);
        attr[dev] = true;
    }
    Gemm4hArgs a;
    a.A = d->A;
    a.W = d->W;
    a.Y = d->Y;
    a.bias = d->bias ? d->bias : (const float*)mage_zero_page();
    a.ln_stats = d->ln_stats;
    a.ln_colsum = d->ln_colsum;
    a.M = d->M;
    a.N = d->N;
    a.lda = d->lda;
    a.ldw = d->ldw ? d->ldw : d->K;
    a.ldy = d->ldy;
    a.a_off = d->a_off;
    a.y_off = d->y_off;
    a.ntiles_n = d->N / 256;
    a.ntiles = (d->M / 256) * a.ntiles_n;
    const int grid = a.ntiles >= n_cu ? n_cu : ((a.ntiles + 7) & ~7);
    hipLaunchKernelGGL((gemm4h_kernel<ACT, LN, HF>), dim3(grid), dim3(256), H_LDS, s, a);
    MAGE_CHECK_LAUNCH("mage_gemm");
    return 1;
}

}  // namespace

// Called by mage_gemm4_try (gemm4.hip) once a product has passed gemm4_kernel's eligibility checks (M, N multiples of 256, plain rows, 32-bit
// lane offsets; NOT its tile-count rule): 1 = launched, 0 = not this kernel's form.  Its form: K = 512 (one epilogue piece per K slab), 16-bit
// rows out, the epilogues act(acc + bias) and its LayerNorm-consuming form -- the decoder's QKV and c_fc in the 16-bit modes.
// Options (mage_set_option): gemm_no_4h = 1: gemm4_kernel runs instead (tests compare the two bitwise); gemm_4h_plain = 1: also the forms
// without QuickGELU (default: only the QuickGELU forms come here -- the others gain 0..5 %, inside the box-to-box spread).
int mage_gemm4h_try(const mage_gemm_desc* d, hipStream_t s, int n_cu) {
    if (mage_options().gemm_no_4h) return 0;
    // Which products (measured, profiles/r06_gemm4h_small_m.txt; one MI355X):
    //   >= 4 tiles per CU (the full-loop sizes): the QuickGELU forms (c_fc: +6-9 %); the others gain 0-5 % there and stay on gemm4_kernel
    //   3/4 .. 4 tiles per CU (the incremental loop's step at 8 k - 32 k rows, where gemm4_kernel loses to the 8-wave kernels: a tile's exposed
    //     epilogue is not amortised): every form -- c_fc +9 % at 16 k rows, +23 % at 8 k; QKV +16 % / +32 %
    //   below 3/4 tile per CU (<= 4 k rows): -20 % against the lockstep kernel: not taken
    const long ntiles = (long)(d->M / 256) * (d->N / 256);
    if (ntiles * 4 < 3L * n_cu) return 0;
    if (ntiles >= 4L * n_cu && d->act != MAGE_ACT_QUICKGELU && !mage_options().gemm_4h_plain) return 0;
    if (d->K != 512 || d->y_dtype != d->dtype || d->y2 || (d->act != MAGE_ACT_NONE && d->act != MAGE_ACT_QUICKGELU)) return 0;
    if (d->ln_stats && (((uintptr_t)d->ln_stats) & 15)) return 0;
    const bool hf = d->dtype == MAGE_F16;
    if (hf) {
        if (d->ln_stats) {
            if (d->act == MAGE_ACT_NONE) return launch4h<MAGE_ACT_NONE, LN_CONSUME, true>(d, s, n_cu);
            return launch4h<MAGE_ACT_QUICKGELU, LN_CONSUME, true>(d, s, n_cu);
        }
        if (d->act == MAGE_ACT_NONE) return launch4h<MAGE_ACT_NONE, LN_NONE, true>(d, s, n_cu);
        return launch4h<MAGE_ACT_QUICKGELU, LN_NONE, true>(d, s, n_cu);
    }
    if (d->ln_stats) {
        if (d->act == MAGE_ACT_NONE) return launch4h<MAGE_ACT_NONE, LN_CONSUME, false>(d, s, n_cu);
        return launch4h<MAGE_ACT_QUICKGELU, LN_CONSUME, false>(d, s, n_cu);
    }
    if (d->act == MAGE_ACT_NONE) return launch4h<MAGE_ACT_NONE, LN_NONE, false>(d, s, n_cu);
    return launch4h<MAGE_ACT_QUICKGELU, LN_NONE, false>(d, s, n_cu);
}
