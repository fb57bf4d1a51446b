// Fused GEMM / implicit-GEMM convolution on CDNA4 matrix cores (see include/mage_hip.h, mage_gemm).
//
// Two kernels share the tile geometry, the LDS image, the tile schedule and the epilogues:
//   gemm_kernel  -- the general one (fp32 and bf16, implicit-GEMM gather, every epilogue kind, any K): lockstep K loop, one
//                   barrier per K slab, described below;
//   gemm8_kernel -- the 8-phase ping-pong variant for the shapes the decoder spends its time in (bf16, plain A, K % 64 == 0,
//                   lean epilogue kinds, >= 2 tiles per CU); its schedule and hazard table are at its definition.
//
// Persistent kernel: one 512-thread workgroup per CU walks a list of 256 (rows of A, "m") x 256 (rows of W, "n") output
// tiles.  8 waves as 2(m) x 4(n); each wave owns a 128x64 sub-tile as 8x4 MFMA 16x16 accumulators (128 fp32 registers).
//
// Why 256x256: measured with ablation builds on the decoder shapes (M=262144, K=512), the L2 -> LDS path alone tops out
// near 17.6 TB/s chip-wide; a 256x128 tile (85 FLOP per staged byte) needs 1.55 k cycles of it per K slab while the
// MFMAs need 1.0 k, and the two overlapped poorly (855 TF).  256x256 is 128 FLOP per staged byte and 24 fragment reads
// per 64 MFMAs instead of 16 per 32.
//
// HBM/L2 -> LDS: global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip) into a 2-stage ring of K slabs (128 bytes per
// row = 64 bf16 / 32 fp32; one stage = 256+256 rows = 64 KiB).  The ring is ONE continuous stream across tiles: while
// the last slab of a tile is multiplied and its epilogue runs, the first slab of the next tile is already in flight.
// The barrier is a raw `s_barrier` (a `__syncthreads()` would drain vmcnt where it stands); the next slab's 8 DMA pieces per
// wave go out in the first half of a slab's phases, between the MFMA batches.
//
// An LDS-DMA writes wave-base + lane*16, so the LDS image is lane-linear: [row][8 chunks of 16 B].  Bank conflicts on
// the fragment reads are removed by an XOR swizzle applied on the *source* address of the DMA (physical chunk p of row r
// holds logical chunk p ^ ((r>>1)&7)) and again on the ds_read_b128.
//
// MFMA operand roles are swapped (A-operand = W rows, B-operand = activation rows) so that each lane ends up with 4
// consecutive output columns n of ONE output row m per accumulator.
//
// fp32 mode uses v_mfma_f32_16x16x4_f32 (exact fp32 fma chain, 1/16 of the bf16 rate); it shares the byte-identical
// LDS image, loader and epilogue with the bf16 path: only the inner MFMA differs.
// This header holds the kernel templates and their launchers (anonymous namespace); gemm.hip instantiates the fp32 / bf16 / split-precision
// forms, gemm_f16.hip the f16 ones (MAGE_F16: the same kernels with the other 16-bit element type -- `DT = MAGE_F16` / `HF = true` below).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_shared.h"

#ifndef MAGE_DMA_PHASES
#define MAGE_DMA_PHASES (MT / 2) // the next slab's DMA pieces go out in the first half of a slab's MT phases (2 per phase): issued
                                // later they are still in flight at the slab's closing vmcnt(0) (+2 % on the 4-GEMM block)
#endif
#define MAGE_U0(ph) ((ph) >= MAGE_DMA_PHASES ? NU : (ph) * NU / MAGE_DMA_PHASES)
#ifndef MAGE_ABL
#define MAGE_ABL 0               // 1 = tuning build: skip the epilogue (main loop only)
#endif

int mage_gemm4_try(const mage_gemm_desc* d, hipStream_t s);    // gemm4.hip: 1 = launched, 0 = not eligible, < 0 = error
int mage_conv3x3_c64_try(const mage_gemm_desc* d, hipStream_t s);    // conv_tile.hip: likewise

namespace {

// MT = 16-row MFMA tiles per wave along m: 8 -> 256x256 workgroup tile (bf16 on large problems), 4 -> 128x256 (fp32, whose
// 8x4 accumulator variant spills, and problems with too few 256-row tiles to fill the chip).
constexpr int BN = 256;
template <int DT, bool GATHER, int MT, int EK, bool SPLIT, int NW> constexpr int ring_stages() {
#ifdef MAGE_GEMM_RING2
    return 2;
#else
    // (round 5: the NARROW tile -- MT = 2, NW = 1: 40 KiB per stage, 16 MFMAs per wave and slab -- takes the 3-stage ring too, gather form
    // included: the f8 VQ-VAE's 64-channel 3x3 convolutions ran one slab per L2 round trip, 14.5 us per 256 x 64 tile of 9 slabs)
    return (DT != MAGE_F32 && !SPLIT && ((!GATHER && MT == 4 && NW == 4) || (MT == 2 && NW == 1))) ? 3 : 2;
#endif
}
// NW = waves side by side along n (each owns 64 columns): 4 -> the 256-column tile; 1 -> the NARROW tile of the lockstep kernel
// (all 8 waves stacked along m, 64 columns): outputs with N <= 128 (the cout/4 bottleneck convolutions of the f8 VQ-VAE, the 8- and
// 16-column heads) waste 3/4 of a 256-column tile's matrix-core work; MT = 2 there (256 x 64 tile, 40 KiB per stage).
// NST = stages of the lockstep kernel's slab ring.  3 for the bf16 128x256 tile (small problems: one to three tiles per CU): its slab
// (0.43 µs of matrix-core work per CU) is shorter than the L2/HBM -> LDS round trip, so with ONE slab in flight the K loop ran at the
// memory latency (1.25 µs per slab measured); with two in flight (counted vmcnt) it does not.  The epilogue's staging windows then
// live in the stage the tile's last slab was read from (free until the next tile's third slab is requested): 144 KiB in all.
template <int MT, int NW = 4, int NST = 2> struct Tile {
    static constexpr int BM = MT * 16 * (8 / NW);
    static constexpr int BNT = 64 * NW;
    static constexpr int A_BYTES = BM * 128;                   // A part of a stage: BM rows x 128 bytes
    static constexpr int STAGE_BYTES = A_BYTES + BNT * 128;    // 64 KiB (MT=8) | 48 KiB (MT=4) | 40 KiB (narrow MT=2)
    static constexpr int RING_BYTES = NST * STAGE_BYTES;       // one workgroup (8 waves, 2 per SIMD) per CU
    // + 4 KiB per wave of epilogue staging (epilogue_lean); 3-stage ring: the staging lives in a consumed stage, 2 KiB per wave remain for
    // the residual rows' way into the accumulator layout (x + Linear(.) on the bf16 stream: 160 KiB in all)
    static constexpr int LDS_BYTES = RING_BYTES + (NST == 2 ? 8 * 4096 : 8 * 2048);
    static constexpr int AU = BM / 64;                         // A units (8 rows x 128 B) per wave per slab
    static constexpr int WU = NW;                              // W units per wave per slab
};

struct GemmArgs {
    mage_gemm_desc d;
    const char* zero;
    int ntiles_n, ntiles;
    int tiles_per_split;                   // tiles_m * ntiles_n: tile index = split * tiles_per_split + tm * ntiles_n + tn
    int stagger_groups, stagger_sleeps;    // start group (li % groups) of an XCD's workgroups after group * sleeps s_sleep(16)
};

template <int DT> struct TT;
template <> struct TT<MAGE_F32> { typedef float elem; static constexpr int CH = 4; };
template <> struct TT<MAGE_BF16> { typedef unsigned short elem; static constexpr int CH = 8; };
template <> struct TT<MAGE_F16> { typedef f16_t elem; static constexpr int CH = 8; };
// the 16-bit element type of a kernel: f16 for DT = MAGE_F16 / HF, bf16 otherwise (also what an fp32 GEMM writes as its 16-bit output)
template <bool HF> using H16T = std::conditional_t<HF, f16_t, unsigned short>;



// Epilogue of one 32x64 block of a wave's sub-tile.  The MFMA layout leaves each lane with 4 consecutive n of 16 different rows
// ---- epilogue --------------------------------------------------------------------------------------------------------
// The MFMA (operands swapped) leaves lane (l15 = lane&15, grp = lane>>4) of accumulator [mt][nt] with output row
// mt*16 + l15 and the 4 consecutive columns nt*16 + grp*4 + {0..3}.  One v_permlane16_swap per register between the
// accumulators nt = 2k and 2k+1 gives every lane 8 CONSECUTIVE columns of its row
//     columns 16*(2k + (grp&1)) + 8*(grp>>1) + {0..7}      (k = 0, 1)
// so the epilogue runs and stores straight from registers: no LDS transpose, no barrier, no wait chains (the earlier
// LDS-staged version spent ~10 k cycles per 256x256 tile in exposed ds_write -> ds_read latency), 16-byte bf16 stores.
// Per-column bias is fetched once per tile at the START of its K loop (lands under the MFMAs); BatchNorm scale/shift
// (VQ-VAE convolutions only) are fetched in the epilogue.
struct ColVecs {
    f32x4 bias[2][2];                  // [k][half]: 8 columns per k
};
__device__ __forceinline__ int epi_col(int n0, int k, int lane) { return n0 + 16 * (2 * k + ((lane >> 4) & 1)) + 8 * (lane >> 5); }
__device__ __forceinline__ void load_colvecs(ColVecs& cv, const mage_gemm_desc& d, int n0, int lane) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int n = epi_col(n0, k, lane);
        const int n_ld = n < d.N ? n : 0;               // clamped, never predicated (see epilogue_wave); N % 8 == 0
#pragma unroll
        for (int h = 0; h < 2; ++h) cv.bias[k][h] = d.bias ? *(const f32x4*)(d.bias + n_ld + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}


template <int ACT, typename OT, int MT, int EK>
__device__ __forceinline__ void epilogue_wave(const mage_gemm_desc& d, const ColVecs& cv, f32x4 (&acc)[MT][4], int m0, int n0,
                                              int lane, int plane, long ysplit) {
    const int l15 = lane & 15;
    const bool simple_rows = d.out_h == 1 && d.out_w >= d.M;       // no regrouping: yrow = m*y_mul_x + y_off
    const float lo = d.post_relu ? 0.f : -INFINITY;                 // post-ReLU as one max
    int ncol[2], nld[2];
    bool nv[2];
    f32x4 scale4[2][2], shift4[2][2];
    constexpr bool GEN = EK == EK_GENERAL;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        ncol[k] = epi_col(n0, k, lane);
        nv[k] = ncol[k] < d.N;                                      // N % 8 == 0 (host check): a chunk is all in or all out
        // Columns/rows outside the problem are CLAMPED to valid ones for the loads (hipcc turns a predicated load into a
        // branch + s_waitcnt vmcnt(0) per element, serialising the round trips); only the stores are predicated.
        nld[k] = nv[k] ? ncol[k] : 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            scale4[k][h] = GEN && d.scale ? *(const f32x4*)(d.scale + nld[k] + 4 * h) : f32x4{1.f, 1.f, 1.f, 1.f};
            shift4[k][h] = GEN && d.scale ? *(const f32x4*)(d.shift + nld[k] + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // One output row of the lane per round.  Keeping a second row's reads in flight (tried: 1-ahead prefetch, rows in
    // pairs) costs 16 more live registers next to the 128 accumulators and hipcc answers with 65-175 spilled VGPRs
    // (603 TF instead of 693): measured, reverted.
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int mq = m0 + mt * 16 + l15;
        const int m = min(mq, d.M - 1);
        int yrow;
        long rrow;                                      // row of the residual: the output row, or (res_half) its half-resolution pixel
        if (simple_rows) {
            yrow = m * d.y_mul_x + d.y_off;
            rrow = yrow;
        } else {
            const int img = m / plane;
            const int rem = m - img * plane;
            const int oy = rem / d.out_w;
            const int ox = rem - oy * d.out_w;
            yrow = img * d.y_img_stride + oy * d.y_mul_y + ox * d.y_mul_x + d.y_off;
            rrow = d.res_half ? (long)img * (plane >> 2) + (long)(oy >> 1) * (d.out_w >> 1) + (ox >> 1) : (long)yrow;
        }
        // everything added after the activation (residual + row table), requested before this row's first store
        f32x4 extra[2][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            extra[k][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            extra[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (GEN && d.residual) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (d.res_dtype == MAGE_F32) {
                    const float* rp = (const float*)d.residual + rrow * d.ldr + nld[k];
                    extra[k][0] = *(const f32x4*)rp;
                    extra[k][1] = *(const f32x4*)(rp + 4);
                } else {
                    const uint4 r = *(const uint4*)((const unsigned short*)d.residual + rrow * d.ldr + nld[k]);
                    if (d.res_dtype == MAGE_F16) {
                        extra[k][0] = widen4<f16_t>(uint2{r.x, r.y});
                        extra[k][1] = widen4<f16_t>(uint2{r.z, r.w});
                    } else {
                        extra[k][0] = widen4<unsigned short>(uint2{r.x, r.y});
                        extra[k][1] = widen4<unsigned short>(uint2{r.z, r.w});
                    }
                }
            }
        }
        if (GEN && d.rowadd) {
            const float* tp = d.rowadd + (long)((yrow / d.rowadd_div) % d.rowadd_mod) * d.N;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                extra[k][0] += *(const f32x4*)(tp + nld[k]);
                extra[k][1] += *(const f32x4*)(tp + nld[k] + 4);
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f32x4 v[2];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[mt][2 * k][e]), __float_as_uint(acc[mt][2 * k + 1][e]),
                                                                false, false);
                v[0][e] = __uint_as_float(r[0]);
                v[1][e] = __uint_as_float(r[1]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v[h] = v[h] + cv.bias[k][h];
                if (GEN) v[h] = v[h] * scale4[k][h] + shift4[k][h];
                if (ACT != MAGE_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[h][e] = act_apply<ACT>(v[h][e]);
                }
                if (GEN) v[h] += extra[k][h];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[h][e] = fmaxf(v[h][e], lo);
            }
            // streaming (non-temporal) stores: the output is not re-read by this kernel, keep the XCD's L2 for the
            // activation panels and W that the neighbouring workgroups re-read
            if (mq < d.M && nv[k]) {
                OT* yp = (OT*)d.Y + ysplit + (long)yrow * d.ldy + ncol[k];
                if (sizeof(OT) == 4) {
                    __builtin_nontemporal_store(v[0], (f32x4*)yp);
                    __builtin_nontemporal_store(v[1], (f32x4*)yp + 1);
                } else {
                    u32x4 pk = {pack16x2<OT>(v[0][0], v[0][1]), pack16x2<OT>(v[0][2], v[0][3]), pack16x2<OT>(v[1][0], v[1][1]),
                                pack16x2<OT>(v[1][2], v[1][3])};
                    __builtin_nontemporal_store(pk, (u32x4*)yp);
                }
            }
        }
    }
}

#ifdef MAGE_PROBE
// tuning build: s_memtime stamps of wave 0 per (workgroup, tile): [0] K loop done, [1] epilogue issued, [2] first-slab wait
// done, [3] first-slab barrier passed (tools/gemm_phase_probe.py)
__device__ unsigned long long mage_probe_buf[256 * 64 * 8];   // [0..3] shader clock (per CU), [4..7] 100 MHz wall clock (chip-wide)
__device__ unsigned long long mage_probe_wave[256 * 16 * 8 * 2];  // per (workgroup, tile < 16, wave): epilogue begin / end, shader clock
#define MAGE_WSTAMP(it, p)                                                                                 \
    do {                                                                                                   \
        if (lane == 0 && (it) < 16 && blockIdx.x < 256)                                                    \
            mage_probe_wave[((blockIdx.x * 16 + (it)) * 8 + wave) * 2 + (p)] = __builtin_readcyclecounter(); \
    } while (0)
__device__ unsigned long long mage_probe_seg[8 * 160];   // gemm8: workgroup 8, one tile: per wave, 4 stamps per phase
#define MAGE_SEG(i)                                                                                        \
    do {                                                                                                   \
        if (lane == 0 && blockIdx.x == 8 && seg_on && (i) < 160) mage_probe_seg[wave * 160 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#define MAGE_STAMP(it, p)                                                                                  \
    do {                                                                                                   \
        if (tid == 0 && (it) < 64 && blockIdx.x < 256) {                                                   \
            mage_probe_buf[(blockIdx.x * 64 + (it)) * 8 + (p)] = __builtin_readcyclecounter();             \
            mage_probe_buf[(blockIdx.x * 64 + (it)) * 8 + 4 + (p)] = __builtin_amdgcn_s_memrealtime();     \
        }                                                                                                  \
    } while (0)
#else
#define MAGE_STAMP(it, p)
#define MAGE_WSTAMP(it, p)
#define MAGE_SEG(i)
#endif




// SPLIT: the split-K form (mage_gemm_desc::n_split > 1).  A template parameter so that the kernels of the generation path keep
// their exact code (the tile decode, two 64-bit strides and the W row stride cost the 8-phase kernel 11 spilled SGPRs otherwise).
// SPL: split-precision operands (1 = bf16 pieces, 2 = f16 pieces; DT = MAGE_BF16 geometry: 128-byte slabs of 64 16-bit elements).  A and W
// rows are [hi(64) | lo(64)] per 64-column slab of K; the K loop runs 3 * K/64 slabs: first, per logical slab, (A_hi, W_lo) then (A_lo, W_hi)
// -- the small terms -- then (f16: accumulators * 2^-11, undoing the scale the lo pieces are stored with) the K/64 (A_hi, W_hi) slabs.
// Only the DMA source offset of a slab, the slab count and the MFMA opcode differ from the bf16 kernel.
template <int DT, bool GATHER, int ACT, int MT, int EK, bool SPLIT = false, int LN = LN_NONE, int NW = 4, int SPL = 0, bool RB = false>
__global__ __launch_bounds__(512, 2) void gemm_kernel(const GemmArgs g) {
    static_assert(!RB || (DT != MAGE_F32 && EK == EK_RES_INIT && SPL == 0), "16-bit residual stream: the bf16 / f16 x + Linear(.) kinds");
    static_assert(SPL == 0 || (DT == MAGE_BF16 && !GATHER && !SPLIT && LN == LN_NONE && EK != EK_GENERAL), "split-precision form: plain bf16-geometry GEMM, lean epilogues");
    typedef typename TT<DT>::elem E;
    typedef H16T<DT == MAGE_F16> H16;                  // 16-bit output / residual rows
    constexpr int NST = ring_stages<DT, GATHER, MT, EK, SPLIT, NW>();
    typedef Tile<MT, NW, NST> TL;
    constexpr int BM = TL::BM, A_BYTES = TL::A_BYTES, STAGE_BYTES = TL::STAGE_BYTES, AU = TL::AU, WU = TL::WU, BNT = TL::BNT;
    constexpr int CH = TT<DT>::CH;
    constexpr int BK = 8 * CH;
    constexpr int ES = sizeof(E);
    const mage_gemm_desc& d = g.d;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // mage_gemm_desc::a_relu: the 256 x 64 tile's plain bf16 form only (host: launch_ek)
    constexpr bool ARELU = DT == MAGE_BF16 && !GATHER && MT == 2 && NW == 1 && EK == EK_BIAS && LN == LN_NONE && !SPLIT && SPL == 0;
    [[maybe_unused]] const bool a_relu = ARELU && d.a_relu != 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- persistent tile schedule.  Workgroup b runs on XCD b%8 (observed dispatch order; only speed depends on it):
    // each XCD owns a contiguous chunk of the tile list and its 32 workgroups walk it side by side, so the n-tiles that
    // share an activation panel, and the whole W matrix, stay in that XCD's L2.
    const int nwg8 = gridDim.x >> 3;                   // workgroups per XCD (grid is a multiple of 8)
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int q8 = g.ntiles >> 3, r8 = g.ntiles & 7;
    const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int chunk1 = chunk0 + q8 + (xcd < r8 ? 1 : 0);
    [[maybe_unused]] const int nk2 = SPL ? 2 * (d.K >> 6) : 0;     // SPL: slabs of the two small-term passes (K % 64 == 0, host check)
    const int nk = SPL ? 3 * (d.K >> 6) : (d.K + BK - 1) / BK;
    __builtin_assume(nk > 0);                          // K > 0 (host check): lets hipcc see that the K loop's vmcnt(0) always runs
    const int plane = d.out_h * d.out_w;

    // ---- loader: each wave moves AU A units + 4 W units (a unit = 8 rows x 128 B = one wave-wide DMA) per slab
    const int lr = lane >> 3;            // row inside a unit
    const int lp = lane & 7;             // physical 16-byte chunk
    const char* a_row[AU];               // plain mode: row base pointer (or null)
    int a_img[AU], a_iy[AU], a_ix[AU];   // gather mode
    const char* w_row[WU];
    int acs[AU], wcs[WU];                // logical chunk this lane fetches for each unit
    int ld_tile = chunk0 + li, ld_kt = 0, ld_stage = 0;
    // GATHER with cin a multiple of the slab width (every 64-channel convolution): a slab lies in ONE tap, so the tap of the loader's slab is a
    // wave-uniform cursor (ky, kx, first channel) stepped once per slab -- no per-lane divisions in the load path (two integer divisions per
    // DMA unit and lane made the f8 VQ-VAE's narrow 3x3 convolutions loader-bound: 16 MFMAs of work per ~250 VALU instructions)
    [[maybe_unused]] const bool tap_uniform = GATHER && d.cin % BK == 0;
    [[maybe_unused]] int ld_ky = 0, ld_kx = 0, ld_c0 = 0;

    auto loader_set_tile = [&](int tile) {
        const int ts = SPLIT ? tile / g.tiles_per_split : 0, trem = SPLIT ? tile - ts * g.tiles_per_split : tile;   // split-K slice
        const int tm = trem / g.ntiles_n, tn = trem - tm * g.ntiles_n;
        const long a_sp = SPLIT ? (long)ts * d.a_split_stride * ES : 0, w_sp = SPLIT ? (long)ts * d.w_split_stride * ES : 0;
#pragma unroll
        for (int i = 0; i < AU; ++i) {
            const int r = (wave * AU + i) * 8 + lr;
            acs[i] = lp ^ ((r >> 1) & 7);
            const int m = tm * BM + r;
            const bool mv = m < d.M;
            const int img = m / plane;
            const int rem = m - img * plane;
            const int oy = rem / d.out_w;
            const int ox = rem - oy * d.out_w;
            if (GATHER) {
                a_img[i] = mv ? img * d.a_img_stride + d.a_off : -1;
                a_iy[i] = oy * d.stride + d.dy0;
                a_ix[i] = ox * d.stride + d.dx0;
            } else {
                const long arow = (long)img * d.a_img_stride + (long)oy * d.in_w + ox + d.a_off;
                a_row[i] = mv ? (const char*)d.A + arow * d.lda * ES + a_sp : nullptr;
            }
        }
#pragma unroll
        for (int i = 0; i < WU; ++i) {
            const int r = (wave * WU + i) * 8 + lr;
            wcs[i] = lp ^ ((r >> 1) & 7);
            const int n = tn * BNT + r;
            w_row[i] = (n < d.N) ? (const char*)d.W + (long)n * ((SPLIT || SPL) ? d.ldw : d.K) * ES + w_sp : nullptr;
        }
    };

    // DMA of this wave's unit u (A units 0..AU-1, then W units) of slab (ld_tile, ld_kt) into stage ld_stage
    auto issue_one = [&](int u, bool live = true) {
        char* sa = smem + ld_stage * STAGE_BYTES;
        // SPL: physical 128-byte unit of this slab along the split row: small-term passes walk the row in order for A (hi, lo, hi, lo ..)
        // and pairwise swapped for W (lo, hi, ..); the main pass takes every hi unit
        int kta = ld_kt, ktw = ld_kt;
        if constexpr (SPL != 0) {
            const bool low = ld_kt < nk2;
            kta = low ? ld_kt : 2 * (ld_kt - nk2);
            ktw = low ? (ld_kt ^ 1) : kta;
        }
        if (u < AU) {
            const int i = u;
            const int kc = kta * BK + acs[i] * CH;
            const char* src = g.zero;
            if (GATHER) {
                if (live && kc < d.K && a_img[i] >= 0) {
                    int ci, ky, kx;
                    if (tap_uniform) {
                        ci = ld_c0 + acs[i] * CH;
                        ky = ld_ky;
                        kx = ld_kx;
                    } else {
                        const int tap = kc / d.cin;
                        ci = kc - tap * d.cin;
                        ky = tap / d.taps_w;
                        kx = tap - ky * d.taps_w;
                    }
                    const int iy = a_iy[i] + ky * d.dys;
                    const int ix = a_ix[i] + kx * d.dxs;
                    if ((unsigned)iy < (unsigned)d.in_h && (unsigned)ix < (unsigned)d.in_w)
                        src = (const char*)d.A + ((long)(a_img[i] + (d.a_half ? (iy >> 1) * (d.in_w >> 1) + (ix >> 1) : iy * d.in_w + ix)) * d.lda + ci) * ES;
                }
            } else {
                src = (live && (SPL || kc < d.K) && a_row[i]) ? a_row[i] + (long)kc * ES : g.zero;
            }
            MAGE_DASSERT(ld_stage >= 0 && ld_stage < NST && (wave * AU + i) * 1024 + 1024 <= A_BYTES);
            glds16(src, sa + (wave * AU + i) * 1024);
        } else {
            const int i = u - AU;
            const int kc = ktw * BK + wcs[i] * CH;
            const char* wsrc = (live && (SPL || kc < d.K) && w_row[i]) ? w_row[i] + (long)kc * ES : g.zero;
            MAGE_DASSERT(A_BYTES + (wave * WU + i) * 1024 + 1024 <= STAGE_BYTES);
            glds16(wsrc, sa + A_BYTES + (wave * WU + i) * 1024);
        }
    };
    auto issue_all = [&]() {
#pragma unroll
        for (int u = 0; u < AU + WU; ++u) issue_one(u);
    };
    auto loader_advance = [&]() {
        ld_stage = ld_stage + 1 == NST ? 0 : ld_stage + 1;
        if constexpr (GATHER) {                        // the uniform tap cursor: next 64 channels, then the next tap
            ld_c0 += BK;
            if (ld_c0 >= d.cin) {
                ld_c0 = 0;
                if (++ld_kx == d.taps_w) {
                    ld_kx = 0;
                    ++ld_ky;
                }
            }
        }
        if (++ld_kt == nk) {
            ld_kt = 0;
            if constexpr (GATHER) ld_ky = ld_kx = ld_c0 = 0;
            ld_tile += nwg8;
            if (ld_tile < chunk1) loader_set_tile(ld_tile);
        }
    };

    // ---- compute state: wave (wm, wn) owns rows [wm*MT*16, +MT*16) x columns [wn*64, +64) of the tile
    const int wm = wave / NW, wn = wave % NW;
    const int l15 = lane & 15, grp = lane >> 4;
    const int rsw = (l15 >> 1) & 7;                    // ((row>>1)&7) for every fragment row of this lane
    const int xoff = (wm * MT * 16 + l15) * 128;       // + mt*16*128
    const int woff = A_BYTES + (wn * 64 + l15) * 128;  // + nt*16*128
    f32x4 acc[MT][4];

    int c_tile = chunk0 + li;
    if (c_tile >= chunk1) return;                      // more workgroups than tiles in this XCD's chunk
    loader_set_tile(ld_tile);
    // Staggered start.  Every workgroup runs the same K loop, so left alone all 256 reach their epilogues together: the
    // chip alternates between "nobody touches HBM" and one burst of 256 x (residual tile in + output tile out) that runs at
    // the HBM write/read limit (5.2 TB/s measured) while every matrix core idles.  Starting the workgroups of an XCD in
    // groups a fraction of a tile period apart spreads the bursts under the other groups' K loops (host: launch_tile).
    if (g.stagger_groups > 1) {
        for (int w = (li % g.stagger_groups) * g.stagger_sleeps; w > 0; --w) __builtin_amdgcn_s_sleep(16);   // 1024 clocks each
    }
    issue_all();
    loader_advance();
    if constexpr (NST == 3) {                          // two slabs in flight from here on
        const bool more = ld_tile < chunk1;
#pragma unroll
        for (int u = 0; u < AU + WU; ++u) issue_one(u, more);
        if (more) loader_advance();
    }
    int c_stage = 0;

    for (int it = 0; c_tile < chunk1; c_tile += nwg8, ++it) {
        const int ts = SPLIT ? c_tile / g.tiles_per_split : 0, trem = SPLIT ? c_tile - ts * g.tiles_per_split : c_tile;
        const int tm = trem / g.ntiles_n, tn = trem - tm * g.ntiles_n;
        MAGE_DASSERT(c_tile >= 0 && c_tile < g.ntiles && tm * BM < d.M && tn * BNT < d.N);
        const long ysplit = SPLIT ? (long)ts * d.y_split_stride : 0;
        const int m0 = tm * BM + wm * MT * 16, n0 = tn * BNT + wn * 64;
        // RB (16-bit residual stream, 16-bit rows out: host check): the residual is added in the epilogue (epilogue_lean RESE) and the
        // accumulators start at 0; an fp32 residual seeds the accumulators before the K loop
        constexpr bool res_epi = RB;
        if constexpr (EK == EK_RES_INIT) {
            // y = x + (A W^T + b): start the accumulators from the fp32 residual.  32 independent 16-byte loads per lane,
            // straight into the MFMA layout (row mt*16 + l15, columns nt*16 + grp*4 + {0..3}), no register cost, one
            // round trip per tile that the first slab's vmcnt(0) below absorbs together with the previous tile's store acks.
            if constexpr (res_epi) {
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    const int m = min(m0 + a * 16 + l15, d.M - 1);
                    const float* rp = (const float*)d.residual + (long)(m * d.y_mul_x + d.y_off) * d.ldr;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int n = n0 + b * 16 + grp * 4;
                        acc[a][b] = *(const f32x4*)(rp + (n < d.N ? n : 0));
                    }
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        ColVecs cv;                                    // EK_GENERAL: bias in the permuted epilogue layout
        f32x4 biasm[4];                                // lean kinds: bias in the MFMA layout (columns nt*16 + grp*4 + {0..3})
        if constexpr (EK == EK_GENERAL) {
            load_colvecs(cv, d, n0, lane);             // lands under the K loop
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int n = n0 + b * 16 + grp * 4;
                biasm[b] = d.bias ? *(const f32x4*)(d.bias + (n < d.N ? n : 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        [[maybe_unused]] LnConsume lnc;                // LN_CONSUME: (mean, rstd) of this lane's rows, s_n of its columns
        if constexpr (LN == LN_CONSUME) {
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const float2 st = *(const float2*)(d.ln_stats + 2 * (long)min(m0 + a * 16 + l15, d.M - 1));
                lnc.mean[a] = st.x;
                lnc.rstd[a] = st.y;
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int n = n0 + b * 16 + grp * 4;
                lnc.s[b] = *(const f32x4*)(d.ln_colsum + (n < d.N ? n : 0));
            }
        }
        auto lo_scale = [&](int kt) __attribute__((always_inline)) {
            if constexpr (SPL == 2) {
                if (kt == nk2) {                       // f16 pieces: the small terms (and the residual) carry the lo pieces' 2^11
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[a][b] *= (1.0f / MAGE_F16_LO_SCALE);
                }
            }
        };
        // one slab, after the wait for its DMAs
        auto slab = [&](int kt) __attribute__((always_inline)) {
            asm volatile("" ::: "memory");
            if constexpr (SPL == 2 && EK == EK_RES_INIT) {
                if (kt == 0) {                         // the residual has landed: give it the lo pieces' scale (exact), undone at kt == nk2
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[a][b] *= MAGE_F16_LO_SCALE;
                }
            }
            if (kt == 0) MAGE_STAMP(it, 2);
            ring_barrier();                            // everyone's share of the slab is in LDS, and every wave is done
            if (kt == 0) MAGE_STAMP(it, 3);
                                                       // reading the other stage, which the DMAs below refill
            const bool more = ld_tile < chunk1;
            const char* st = smem + c_stage * STAGE_BYTES;
            // Software-pipelined phases: MT phases per slab (2 k-halves x MT/2 pairs of 16-row tiles).  Each phase requests
            // the NEXT phase's fragments (2 ds_read_b128, or the 6 that open the second k-half), issues its share of the
            // next slab's DMAs, then runs 8 MFMAs on fragments requested one phase earlier: LDS latency and DMA issue sit
            // under the matrix pipe instead of in front of it (the compiler's own order was read-all / wait / MFMA-all).
            constexpr int NG = MT / 2, NU = AU + WU;
            const char* xs = st + xoff;
            const char* ws = st + woff;
            const int pcs[2] = {((grp + 0) ^ rsw) * 16, ((grp + 4) ^ rsw) * 16};
            u32x4 wf[2][4], xf[2][MT];
#if MAGE_ABL != 6
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[0][i] = *(const u32x4*)(ws + i * 2048 + pcs[0]);
            xf[0][0] = *(const u32x4*)(xs + pcs[0]);
            xf[0][1] = *(const u32x4*)(xs + 2048 + pcs[0]);
#endif
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int gq = 0; gq < NG; ++gq) {
                    const int ph = t * NG + gq;
#if MAGE_ABL != 5
                    // Plain GEMM: ALWAYS issue (the zero page, into the stage nobody reads, once the tile list is exhausted) so
                    // the slab body is one basic block and hipcc counts lgkmcnt exactly instead of draining it at every join.
                    if (GATHER && NST == 2) {
                        if (more) {
#pragma unroll
                            for (int u = MAGE_U0(ph); u < MAGE_U0(ph + 1); ++u) issue_one(u);
                        }
                    } else {                           // (3-stage ring: counted waits need every slab to issue its full set of loads)
#pragma unroll
                        for (int u = MAGE_U0(ph); u < MAGE_U0(ph + 1); ++u) issue_one(u, more);
                    }
#endif
#if MAGE_ABL != 6
                    if (gq + 1 < NG) {
                        xf[t][2 * gq + 2] = *(const u32x4*)(xs + (2 * gq + 2) * 2048 + pcs[t]);
                        xf[t][2 * gq + 3] = *(const u32x4*)(xs + (2 * gq + 3) * 2048 + pcs[t]);
                    } else if (t == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) wf[1][i] = *(const u32x4*)(ws + i * 2048 + pcs[1]);
                        xf[1][0] = *(const u32x4*)(xs + pcs[1]);
                        xf[1][1] = *(const u32x4*)(xs + 2048 + pcs[1]);
                    }
                    if constexpr (ARELU) {             // the product over relu(A): max(., 0) on the fragments' 16-bit lanes as integers
                        if (a_relu) {
#pragma unroll
                            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                                for (int q = 0; q < 4; ++q) xf[t][2 * gq + mm][q] = relu16x2(xf[t][2 * gq + mm][q]);
                        }
                    }
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mm = 0; mm < 2; ++mm) {
                            const int mt = 2 * gq + mm;
                            if constexpr (SPL == 2) {
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                    __builtin_bit_cast(f16x8, wf[t][nt]), __builtin_bit_cast(f16x8, xf[t][mt]), acc[mt][nt], 0, 0, 0);
                            } else if constexpr (DT != MAGE_F32) {
                                acc[mt][nt] = mfma16x16x32<E>(wf[t][nt], xf[t][mt], acc[mt][nt]);
                            } else {
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj)
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                        __uint_as_float(wf[t][nt][jj]), __uint_as_float(xf[t][mt][jj]), acc[mt][nt], 0, 0, 0);
                            }
                        }
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
            }
            if (more) loader_advance();
            c_stage = c_stage + 1 == NST ? 0 : c_stage + 1;
        };
        // The slab to multiply was issued one whole iteration (or one epilogue) ago; with the 2-stage ring nothing younger is in
        // flight: vmcnt(0).  (the builtin, not inline asm: hipcc's own wait-count pass must SEE this wait, or it guards every later
        // use of the bias vectors fetched above with its own vmcnt(0) — in the epilogue that meant "wait for the previous row's
        // store ack" 16 times per tile.)  3-stage ring: the tile's first slab still waits for everything (residual tile, bias, the
        // previous tile's store acks; the second slab was requested long ago), the later ones leave the AU + WU DMAs of the slab
        // after them in flight -- peeled so that the vmcnt(0) dominates every later use of the tile-start loads.
        if constexpr (NST == 2) {
            for (int kt = 0; kt < nk; ++kt) {
                lo_scale(kt);
                __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0), expcnt/lgkmcnt untouched
                slab(kt);
            }
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);
            slab(0);
            for (int kt = 1; kt < nk; ++kt) {
                lo_scale(kt);
                __builtin_amdgcn_s_waitcnt(0x0F70 | (AU + WU));      // vmcnt(AU + WU <= 15)
                slab(kt);
            }
        }
#if MAGE_ABL == 1 || MAGE_ABL == 5 || MAGE_ABL == 6
        {   // tuning build: main loop only (keep the accumulators alive, store nothing)
            float sacc = 0.f;
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) sacc += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
            if (sacc == 123456.789f) ((float*)d.Y)[0] = sacc;
            continue;
        }
#endif
        // ---- epilogue: straight from the accumulators (no LDS), while the next tile's first slab lands in the other stage
        MAGE_STAMP(it, 0);
        MAGE_WSTAMP(it, 0);
        if constexpr (EK == EK_GENERAL) {
            if (d.y_dtype == MAGE_F32) epilogue_wave<ACT, float, MT, EK>(d, cv, acc, m0, n0, lane, plane, ysplit);
            else epilogue_wave<ACT, H16, MT, EK>(d, cv, acc, m0, n0, lane, plane, ysplit);
        } else {
            char* stg = smem + TL::RING_BYTES + wave * 4096;
            if constexpr (NST == 3) {                  // staging in the stage of the tile's last slab, once every wave has read it
                ring_barrier();
                stg = smem + (c_stage == 0 ? NST - 1 : c_stage - 1) * STAGE_BYTES + wave * 4096;
            }
            if constexpr (LN == LN_CONSUME) {
                if (d.y_dtype == MAGE_F32) epilogue_lean<ACT, float, MT, false, LN>(d, biasm, acc, m0, n0, lane, plane, stg, ysplit, &lnc);
                else epilogue_lean<ACT, H16, MT, false, LN>(d, biasm, acc, m0, n0, lane, plane, stg, ysplit, &lnc);
            } else if constexpr (LN == LN_PRODUCE) {
                // fp32 stream + 16-bit copy, or (16-bit y_dtype) the 16-bit stream alone
                if (d.y_dtype == MAGE_F32) epilogue_lean<ACT, float, MT, false, LN, 0, H16>(d, biasm, acc, m0, n0, lane, plane, stg, ysplit);
                else epilogue_lean<ACT, H16, MT, false, LN, 0, H16, RB>(d, biasm, acc, m0, n0, lane, plane, stg, ysplit);
            } else if constexpr (LN == LN_DUAL || LN == LN_GELUBWD) {
                epilogue_lean<ACT, unsigned short, MT, false, LN>(d, biasm, acc, m0, n0, lane, plane, stg, ysplit);           // bf16 rows (host check)
            } else if constexpr (SPL != 0) {
                if (d.y_dtype == MAGE_F32) epilogue_lean<ACT, float, MT>(d, biasm, acc, m0, n0, lane, plane, stg, ysplit);
                else epilogue_lean<ACT, float, MT, false, LN_NONE, SPL>(d, biasm, acc, m0, n0, lane, plane, stg, ysplit);   // split rows out
            } else {
                if (d.y_dtype == MAGE_F32) epilogue_lean<ACT, float, MT>(d, biasm, acc, m0, n0, lane, plane, stg, ysplit);
                else epilogue_lean<ACT, H16, MT, false, LN_NONE, 0, H16, RB>(d, biasm, acc, m0, n0, lane, plane, stg, ysplit);
            }
        }
        MAGE_STAMP(it, 1);
        MAGE_WSTAMP(it, 1);
    }
}

// LN_HEAD epilogue of the 8-phase padded-taps kernel (gemm_shared.h): the tile's rows y = act(acc + bias), rounded to bf16 as a store would
// round them, never leave the CU; head[row][t] = sum_n y[row][n] head_w[t][n] (t < 16) does.  Wave (wr, wc) holds rows wr*128 + [0,128) x
// columns wc*64 + [0,64) in the MFMA output layout: lane (l15, grp) has row l15 of each 16-row tile, columns nt*16 + grp*4 + {0..3}.  Packed
// to bf16, two 16-column blocks (nt = 2t, 2t+1) ARE a B operand of v_mfma_f32_16x16x32_bf16 for row l15 -- 8 values of "k" per lane; which
// channel sits at which k position only has to agree with the A operand, so the head weights are fetched in that order (hw[t]: tap l15,
// channels n0 + 32t + {grp*4 + 0..3, 16 + grp*4 + 0..3}).  16 MFMAs per wave and tile give the wave's 64-channel partial sums; the four
// wave columns are added in the fixed order wc = 0..3 through the staging windows (4 KiB per wave = 64 rows x 16 taps: two halves), each
// wave finishing 16 of the 64 rows.  All eight waves run this in step (the K loop's half-offset is closed before the epilogue).
// With d.residual (bf16 rows [.., N], optionally at half resolution: res_half) the rows are y = act((acc + bias) + residual) -- a bottleneck
// block's last convolution + its identity path + the ReLU that follows the block (vqvae_model.py:147-166,210), fetched in the accumulator
// layout one 16-row tile ahead (8 bytes per lane and 16-column block: the identity path is a quarter-resolution tensor that lives in L2).
template <int ACT>
__device__ __forceinline__ void epilogue_head(const mage_gemm_desc& d, const f32x4 (&bias)[4], f32x4 (&acc)[8][4], const u32x4 (&hw)[2],
                                              int tile_m0, int lane, int wave, int plane, char* stg_all, long phase_rows) {
    const int l15 = lane & 15, grp = lane >> 4, wr = wave >> 2, wc = wave & 3;
    f32x4 h[8];
    const bool has_res = d.residual != nullptr;        // uniform
    uint2 rres[2][4];
    auto res_request = [&](int a) {
        const int m = tile_m0 + wr * 128 + a * 16 + l15;
        const int img = m / plane, rem = m - img * plane;
        const int oy = rem / d.out_w, ox = rem - oy * d.out_w;
        const long rrow = d.res_half ? (long)img * (plane >> 2) + (long)(oy >> 1) * (d.out_w >> 1) + (ox >> 1) : (long)m;
        const unsigned short* rp = (const unsigned short*)d.residual + rrow * d.ldr + wc * 64 + grp * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) rres[a & 1][q] = *(const uint2*)(rp + q * 16);
    };
    if (has_res) res_request(0);
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        h[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_res && a + 1 < 8) res_request(a + 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 v0 = acc[a][2 * t] + bias[2 * t], v1 = acc[a][2 * t + 1] + bias[2 * t + 1];
            if (has_res) {
                v0 += widen4<unsigned short>(rres[a & 1][2 * t]);
                v1 += widen4<unsigned short>(rres[a & 1][2 * t + 1]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v0[e] = act_apply<ACT>(v0[e]);
                v1[e] = act_apply<ACT>(v1[e]);
            }
            const u32x4 y = u32x4{pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
            h[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hw[t]), __builtin_bit_cast(bf16x8, y), h[a], 0, 0, 0);
        }
    }
    // h[a] = taps grp*4 + {0..3} of row a*16 + l15, summed over this wave's 64 channels
    char* mine = stg_all + wave * 4096;
    const char* col0 = stg_all + wr * 4 * 4096;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(mine + (i * 16 + l15) * 64 + grp * 16) = h[half * 4 + i];
        __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): the partial sums are in LDS before the barrier says so
        ring_barrier();
        const int r = wc * 16 + l15;                   // this wave finishes rows [wc*16, wc*16 + 16) of the half
        f32x4 s = *(const f32x4*)(col0 + r * 64 + grp * 16);
#pragma unroll
        for (int w = 1; w < 4; ++w) s += *(const f32x4*)(col0 + w * 4096 + r * 64 + grp * 16);
        const int m = tile_m0 + wr * 128 + half * 64 + r;
        const int img = m / plane, rem = m - img * plane;
        const int oy = rem / d.out_w, ox = rem - oy * d.out_w;
        const long yrow = (long)img * d.y_img_stride + (long)oy * d.y_mul_y + (long)ox * d.y_mul_x + d.y_off + phase_rows;
        // (ldy == 4: a head with at most 4 outputs -- the f8 decoder's RGB head: only taps 0..3 leave the CU, 16 bytes per row)
        if (d.ldy >= 16 || grp == 0) __builtin_nontemporal_store(s, (f32x4*)((float*)d.Y + yrow * d.ldy + grp * 4));
        if (half == 0) {
            __builtin_amdgcn_s_waitcnt(0xC07F);        // the reads of half 0 are done before anyone overwrites a window
            ring_barrier();
        }
    }
}

// ======================================================================================================================
// gemm8_kernel: the 8-phase ping-pong variant of the bf16 256x256 plain GEMM with a lean epilogue (the decoder's Linear
// layers).  Structure after cdna_hip_programming.md "The 256^2 8-phase template": the wave's 128x64 output is four 64x32
// quadrants; a PHASE = [12/4/8/4 ds_read_b128 of one quadrant's operand fragments + one 16 KiB "piece" of DMA (2 per wave)]
// s_barrier [16 MFMA] s_barrier; the two wave halves (wr = 0: rows 0-127, wr = 1: rows 128-255; SIMD partners) run ONE
// barrier apart, so on every SIMD one wave is in its 16-MFMA section while its partner reads LDS and issues DMA.  vmcnt is
// never drained in the loop: one counted wait per K slab.
//
// LDS: 2 slab buffers x 4 pieces x 16 KiB (+ 32 KiB epilogue staging).  A piece is 128 rows x 128 B, cut by QUADRANT INDEX,
// not by tile half, so that every piece has a long window between its last read and the first read of its replacement:
//     A_a (a = 0,1): tile rows {wr*128 + a*64 + [0,64)} of both wave halves        read in phase 1 (a=0) / phase 3 (a=1)
//     W_b (b = 0,1): W rows   {wc*64 + b*32 + [0,32)} of the four wave columns     read in phases 1 and 4 (b=0) / 2 (b=1)
// Quadrant order (0,0) (0,1) (1,1) (1,0).  With slab j computed from buffer j&1, phase p of slab j issues
//     p=1: A_1 of slab j+1   p=2: W_0 of slab j+1   p=3: A_0 of slab j+2   p=4: W_1 of slab j+2, then s_waitcnt vmcnt(4)
// WAR (a buffer is restaged >= 2 phases after its last ds_read, which covers the half that runs a barrier behind):
//     A_1 last read j-1.p3 -> restaged j.p1;  W_0: j-1.p4 -> j.p2;  A_0: j.p1 -> j.p3;  W_1: j.p2 -> j.p4.
// RAW (loads return in order, so "at most the 2 newest pieces outstanding" = everything issued up to j.p2 has landed; the
// wait sits before phase 4's first barrier and the data is first read one phase later, after the trailing half has also
// waited): slab j+1's A_0 (issued j-1.p3), W_1 (j-1.p4), A_1 (j.p1), W_0 (j.p2) are all complete at j.p4's wait.
// The loader cursors run across tile boundaries (the next tile's slabs 0 and 1 stream in under this tile's last phases and
// its epilogue).  At a tile's end the leading half gives the trailing half one barrier (both then run the epilogue in
// step), and the trailing half drops back by one barrier before the next tile's first phase.
// TAPS: implicit-GEMM convolution over a ZERO-PADDED input (in_h >= out_h + taps_h - 1, in_w = row pitch >= out_w + taps_w - 1, stride 1): every tap
// of every output pixel is a valid row, so the gather is the plain loader plus ONE scalar offset per K slab -- a slab lies in one tap,
// whose rows sit (ky*in_w + kx) rows further -- kept as scalar cursors per A piece (no vector instruction in a load section, which is
// what this kernel's schedule depends on).  Slab order: bf16 form (channel slab, tap) with the tap fastest (CMAJ below: L2 reuse of the
// padded rows), in every form (the split-precision forms walk both of their passes that way).  The lockstep kernel's generic gather decodes the tap per lane
// and per slab and re-tests the bounds (frame conv3x3: 773 TFLOP/s, 6.7x its algorithmic bytes fetched: round-1 PMC).
// SPL: split-precision operands (see gemm_kernel): 3 * K/64 slabs per tile, the slab -> source offset map in issue(), accumulators scaled
// once between the small-term passes and the main pass (f16 pieces), the MFMA opcode.  Schedule, hazards and LDS image are unchanged.
// HF: f16 operands (MAGE_F16) instead of bf16: the MFMA opcode and the 16-bit conversions of the residual rows / the output differ, nothing else.
template <int ACT, int EK, bool SPLIT = false, bool TAPS = false, int LN = LN_NONE, int SPL = 0, bool RB = false, bool HF = false>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const GemmArgs g) {
    static_assert(SPL == 0 || (!SPLIT && LN == LN_NONE), "split-precision form: no split-K, no LayerNorm fold");
    static_assert(!HF || (SPL == 0 && !SPLIT && LN != LN_HEAD && LN != LN_DUAL && LN != LN_GELUBWD), "f16 form: the generation path's epilogues");
    typedef H16T<HF> H16;
    constexpr int MT = 8, BM = 256;
    constexpr int PIECE = 16384, KBUF = 4 * PIECE;
    constexpr int P_A0 = 0, P_A1 = 1, P_W0 = 2, P_W1 = 3;
    const mage_gemm_desc& d = g.d;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg8 = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int q8 = g.ntiles >> 3, r8 = g.ntiles & 7;
    const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int chunk1 = chunk0 + q8 + (xcd < r8 ? 1 : 0);
    [[maybe_unused]] const int nk2 = SPL ? 2 * (d.K >> 6) : 0;
    const int nk = SPL ? 3 * (d.K >> 6) : (d.K + 63) / 64;
    __builtin_assume(nk > 0);
    const int plane = d.out_h * d.out_w;
    int c_tile = chunk0 + li;
    if (c_tile >= chunk1) return;

    // ---- loader: per piece type a cursor (tile, slab) and this lane's two row pointers (units 2*wave, 2*wave+1 of the piece)
    const int lr = lane >> 3, lp = lane & 7;
    const int lc[2] = {lp ^ ((lr >> 1) & 7), lp ^ ((4 + (lr >> 1)) & 7)};       // logical 16-byte chunk fetched for unit i (swizzle)
    // A piece-unit's source = uniform base (operand + slab offset, SGPRs) + a per-lane 32-bit byte offset (row and swizzled
    // chunk): no vector arithmetic per DMA piece (the partner wave holds priority during its MFMA section: every VALU
    // instruction of a load section waits for a slot).  Rows beyond M / N are clamped to the last valid row (their products
    // only reach outputs that are never stored); the host sends K % 64 != 0 or operands >= 4 GiB to gemm_kernel.
    int cur_tile[4], cur_kt[4], cur_buf[4];
    unsigned voff[4][2];
    auto set_rows = [&](int P) {
        const int tile = cur_tile[P];
        // split-K slice.  The loader cursors run past the end of the tile list (their DMA lands in a buffer nobody reads): rows
        // are clamped below, and so is the slice, or its offset would leave the operand
        const int ts_raw = SPLIT ? tile / g.tiles_per_split : 0, trem = SPLIT ? tile - ts_raw * g.tiles_per_split : tile;
        const int ts = SPLIT ? min(ts_raw, d.n_split - 1) : 0;
        const int tm = trem / g.ntiles_n, tn = trem - tm * g.ntiles_n;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (2 * wave + i) * 8 + lr;                               // row of the piece
            if (P == P_A0 || P == P_A1) {
                const int m = min(tm * BM + (r >> 6) * 128 + (P == P_A1 ? 64 : 0) + (r & 63), d.M - 1);
                const int img = m / plane;
                const int rem = m - img * plane;
                const int oy = rem / d.out_w;
                const int ox = rem - oy * d.out_w;
                // head_phases: column tile tn is sub-pixel phase (py, px) = (tn >> 1, tn & 1); its window starts py rows and px columns further
                const int ph_a = (LN == LN_HEAD && d.head_phases) ? (tn >> 1) * d.in_w + (tn & 1) : 0;
                const long arow = (long)img * d.a_img_stride + (long)oy * d.in_w + ox + d.a_off + ph_a;
                voff[P][i] = (unsigned)(arow * d.lda * 2 + (SPLIT ? (long)ts * d.a_split_stride * 2 : 0) + lc[i] * 16);
            } else {
                const int n = min(tn * BN + (r >> 5) * 64 + (P == P_W1 ? 32 : 0) + (r & 31), d.N - 1);
                voff[P][i] = (unsigned)((long)n * ((SPLIT || SPL) ? d.ldw : d.K) * 2 + (SPLIT ? (long)ts * d.w_split_stride * 2 : 0) + lc[i] * 16);
            }
        }
    };
    // TAPS cursors of the two A piece types: slab inside the tap, kx, and the tap's byte offset ((ky*in_w + kx) rows)
    int tap_ci[2] = {0, 0}, tap_kx[2] = {0, 0};
    long tap_off[2] = {0, 0};
    const int spt = TAPS ? d.cin >> 6 : 1;             // 64-wide slabs per tap
    // CMAJ (bf16 padded-taps form): the K loop walks (channel slab, tap) with the TAP fastest -- a tile's nine (four) visits to the same
    // 128-byte pieces of its padded input rows are then nine consecutive slabs (41 KB of unique input per group at 16 x 16 latents)
    // instead of one visit per 4-slab tap, 36 slabs apart: with 32 workgroups per XCD the tap-major order re-fetched the input 3-4x from
    // beyond L2 (PMC, DESIGN finding 64).  W keeps its documented [N][(ky, kx, ci)] layout: the W pieces read slab tap*spt + c.
    // The split-precision forms walk their two passes (small terms, then main) in the same order: their padded split rows are twice as wide,
    // and the encoder's 3x3 convolutions fetched 3.0-6.3 GB per launch for 0.3-1.1 GB of input (PMC) -- HBM-bound, not matrix-core-bound.
    constexpr bool CMAJ = TAPS;
    [[maybe_unused]] int tap_ky[2] = {0, 0};
    [[maybe_unused]] int w_tap[2] = {0, 0}, w_c[2] = {0, 0}, w_slab[2] = {0, 0};
    [[maybe_unused]] const int ntaps_k = TAPS ? d.taps_h * d.taps_w : 1;
    auto issue = [&](int P) {
        MAGE_DASSERT((cur_buf[P] == 0 || cur_buf[P] == 1) && cur_kt[P] >= 0 && cur_kt[P] < nk);
        char* dst = smem + cur_buf[P] * KBUF + P * PIECE + (2 * wave) * 1024;
        const bool isA = P == P_A0 || P == P_A1;
        const char* sbase;
        // SPL: slab kt of the 3 * K/64: kt < nk2 -> logical slab kt/2, pieces (A hi, W lo) for even kt, (A lo, W hi) for odd; else logical slab
        // kt - nk2, pieces (hi, hi).  The physical 128-byte unit along a split row is 2 * logical + piece.
        [[maybe_unused]] const int kt_ = cur_kt[P];
        [[maybe_unused]] const bool low = kt_ < nk2;
        [[maybe_unused]] const int piece = low ? (isA ? (kt_ & 1) : ((kt_ & 1) ^ 1)) : 0;
        if constexpr (SPL != 0) {
            if (TAPS && isA) sbase = (const char*)d.A + tap_off[P] + (tap_ci[P] * 2 + piece) * 128;
            else if (CMAJ) sbase = (const char*)d.W + (2 * w_slab[P & 1] + piece) * 128;                 // w_slab: the LOGICAL slab tap*spt + c
            else sbase = (const char*)(isA ? d.A : d.W) + ((low ? (kt_ & ~1) : 2 * (kt_ - nk2)) + piece) * 128;
        } else {
            if (TAPS && isA) sbase = (const char*)d.A + tap_off[P] + tap_ci[P] * 128;       // P_A0 = 0, P_A1 = 1 index the cursors
            else if (CMAJ) sbase = (const char*)d.W + w_slab[P & 1] * 128;                  // P_W0 = 2, P_W1 = 3
            else sbase = (const char*)(isA ? d.A : d.W) + cur_kt[P] * 128;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(sbase + voff[P][i], dst + i * 1024);
        cur_buf[P] ^= 1;
        if (CMAJ && (SPL == 0 || !low || (kt_ & 1))) {        // (SPL: the cursors follow the LOGICAL slab)
            if (isA) {                                 // next tap of this channel slab; after the last one, the next slab's first
                tap_off[P] += (long)d.lda * 2;
                if (++tap_kx[P] == d.taps_w) {
                    tap_kx[P] = 0;
                    tap_off[P] += (long)(d.in_w - d.taps_w) * d.lda * 2;
                    if (++tap_ky[P] == d.taps_h) {
                        tap_ky[P] = 0;
                        tap_off[P] = 0;
                        ++tap_ci[P];
                    }
                }
            } else {
                w_slab[P & 1] += spt;
                if (++w_tap[P & 1] == ntaps_k) {
                    w_tap[P & 1] = 0;
                    w_slab[P & 1] = ++w_c[P & 1];
                }
            }
        }
        if (++cur_kt[P] == nk) {
            cur_kt[P] = 0;
            if (TAPS && isA) {
                tap_ci[P] = 0;
                tap_kx[P] = 0;
                tap_off[P] = 0;
                if constexpr (CMAJ) tap_ky[P] = 0;
            }
            if constexpr (CMAJ) {
                if (!isA) {
                    w_tap[P & 1] = 0;
                    w_c[P & 1] = 0;
                    w_slab[P & 1] = 0;
                }
            }
            cur_tile[P] += nwg8;
            set_rows(P);
        } else if (SPL != 0 && TAPS && cur_kt[P] == nk2) {             // the main pass walks the (slab, tap) sequence again from the first
            if (isA) {
                tap_ci[P] = 0;
                tap_kx[P] = 0;
                tap_off[P] = 0;
                tap_ky[P] = 0;
            } else {
                w_tap[P & 1] = 0;
                w_c[P & 1] = 0;
                w_slab[P & 1] = 0;
            }
        }
    };
#pragma unroll
    for (int P = 0; P < 4; ++P) {
        cur_tile[P] = c_tile;
        cur_kt[P] = 0;
        cur_buf[P] = 0;
        set_rows(P);
    }
    if (g.stagger_groups > 1) {
        for (int w = (li % g.stagger_groups) * g.stagger_sleeps; w > 0; --w) __builtin_amdgcn_s_sleep(16);
    }
    // prologue: slab 0 complete, slab 1's A_0 and W_1 (the two pieces the steady state issues two slabs ahead)
    issue(P_A0); issue(P_W1); issue(P_A1); issue(P_W0); issue(P_A0); issue(P_W1);
    __builtin_amdgcn_s_waitcnt(0x0F74);                // vmcnt(4)
    asm volatile("" ::: "memory");
    ring_barrier();

    // ---- compute state
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, grp = lane >> 4;
    const int rsw = (l15 >> 1) & 7;
    const int pc0 = ((grp + 0) ^ rsw) * 16, pc1 = ((grp + 4) ^ rsw) * 16;        // k-half 0 / 1 chunk of this lane's fragment row
    const int a_off = (wr * 64 + l15) * 128;                                     // + m*2048 inside an A piece
    const int w_off = (wc * 32 + l15) * 128;                                     // + n*2048 inside a W piece
    f32x4 acc[MT][4];
    int c_buf = 0;
    [[maybe_unused]] int it = 0;                       // probe build: tile counter of this workgroup

    for (; c_tile < chunk1; c_tile += nwg8, ++it) {
        const int ts = SPLIT ? c_tile / g.tiles_per_split : 0, trem = SPLIT ? c_tile - ts * g.tiles_per_split : c_tile;
        const int tm = trem / g.ntiles_n, tn = trem - tm * g.ntiles_n;
        const long ysplit = SPLIT ? (long)ts * d.y_split_stride : 0;
        MAGE_DASSERT(c_tile >= 0 && c_tile < g.ntiles && tm * BM < d.M && tn * BN < d.N);
        const int m0 = tm * BM + wr * 128, n0 = tn * BN + wc * 64;
        constexpr bool res_epi = RB && !TAPS;         // 16-bit stream, 16-bit rows out (host check): residual added in the epilogue (RESE), acc from 0
        if constexpr (EK == EK_RES_INIT) {
            // y = r + (A W^T + b): the accumulators start from r.  r is the fp32 residual stream, or (rowadd set, residual null) a
            // broadcast row table: r[m] = rowadd[(yrow / div) % mod] -- the H/W positional table of the frame convolution
            // (the table form exists in the TAPS instantiation only: the Linear layers' kernel keeps its exact code)
            if constexpr (res_epi) {
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int m = min(m0 + a * 16 + l15, d.M - 1);
                const float* rp;
                if constexpr (TAPS) {
                    const int img = m / plane, rem = m - img * plane;
                    const int oy = rem / d.out_w, ox = rem - oy * d.out_w;
                    const int yrow = img * d.y_img_stride + oy * d.y_mul_y + ox * d.y_mul_x + d.y_off;       // regrouped rows allowed
                    rp = d.rowadd + (long)((yrow / d.rowadd_div) % d.rowadd_mod) * d.N;
                } else {
                    rp = (const float*)d.residual + (long)(m * d.y_mul_x + d.y_off) * d.ldr;
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int n = n0 + b * 16 + grp * 4;
                    acc[a][b] = *(const f32x4*)(rp + (n < d.N ? n : 0));
                }
            }
            }
        } else {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 biasm[4];
        [[maybe_unused]] LnConsume lnc;                // LN_CONSUME: requested with the bias vector in the tile's last slab
        [[maybe_unused]] u32x4 headw[2];               // LN_HEAD: likewise
        [[maybe_unused]] const bool seg_on = c_tile == chunk0 + li + 2 * nwg8;     // probe build: stamp the third tile
        if (wr) ring_barrier();                        // the trailing half drops one barrier behind
        for (int kt = 0; kt < nk; ++kt) {
            [[maybe_unused]] int seg_ph = kt * 4;
            if constexpr (SPL == 2) {
                if (kt == nk2) {                       // f16 pieces: the small terms (and the residual) carry the lo pieces' 2^11
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[a][b] *= (1.0f / MAGE_F16_LO_SCALE);
                }
            }
            const char* base = smem + c_buf * KBUF;
            u32x4 af[4][2], wf[2][2];                  // [16-row tile of the quadrant][k-half]
            auto read_a = [&](int a) {
                const char* pa = base + (a ? P_A1 : P_A0) * PIECE + a_off;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    af[m][0] = *(const u32x4*)(pa + m * 2048 + pc0);
                    af[m][1] = *(const u32x4*)(pa + m * 2048 + pc1);
                }
            };
            auto read_w = [&](int b) {
                const char* pw = base + (b ? P_W1 : P_W0) * PIECE + w_off;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    wf[n][0] = *(const u32x4*)(pw + n * 2048 + pc0);
                    wf[n][1] = *(const u32x4*)(pw + n * 2048 + pc1);
                }
            };
            auto mfma_quadrant = [&](int a, int b) {
                __builtin_amdgcn_sched_barrier(0);
                MAGE_SEG(seg_ph * 4 + 1);
                ring_barrier();
                __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0)
                MAGE_SEG(seg_ph * 4 + 2);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            if constexpr (SPL == 2)
                                acc[a * 4 + m][b * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                    __builtin_bit_cast(f16x8, wf[n][t]), __builtin_bit_cast(f16x8, af[m][t]), acc[a * 4 + m][b * 2 + n], 0, 0, 0);
                            else
                                acc[a * 4 + m][b * 2 + n] = mfma16x16x32<H16>(wf[n][t], af[m][t], acc[a * 4 + m][b * 2 + n]);
                        }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                MAGE_SEG(seg_ph * 4 + 3);
                ring_barrier();
                __builtin_amdgcn_sched_barrier(0);
                ++seg_ph;
                MAGE_SEG(seg_ph * 4 + 0);
            };
            // phase 1: quadrant (0,0)
            read_w(0);
            __builtin_amdgcn_sched_barrier(0);
            read_a(0);
            issue(P_A1);
            if (EK == EK_RES_INIT && !res_epi && kt == 0) {
                // the residual tile (requested before this slab's two DMA instructions) is in the accumulators: loads return in
                // order, so "at most 2 outstanding" proves it whatever the previous epilogue's stores are doing
                __builtin_amdgcn_s_waitcnt(0x0F72);
                asm volatile("" ::: "memory");
                if constexpr (SPL == 2) {              // give the residual / row table the lo pieces' scale (exact), undone at kt == nk2
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[a][b] *= MAGE_F16_LO_SCALE;
                }
            }
            mfma_quadrant(0, 0);
            // phase 2: quadrant (0,1)
            read_w(1);
            issue(P_W0);
            mfma_quadrant(0, 1);
            // phase 3: quadrant (1,1)
            read_a(1);
            issue(P_A0);
            if (kt == nk - 1) {                        // the epilogue's bias vector, late: its registers are free during the K loop
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int n = n0 + b * 16 + grp * 4;
                    biasm[b] = d.bias ? *(const f32x4*)(d.bias + (n < d.N ? n : 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (LN == LN_CONSUME) lnc.s[b] = *(const f32x4*)(d.ln_colsum + (n < d.N ? n : 0));
                }
                if constexpr (LN == LN_CONSUME) {
#pragma unroll
                    for (int a = 0; a < MT; ++a) {
                        const float2 st = *(const float2*)(d.ln_stats + 2 * (long)min(m0 + a * 16 + l15, d.M - 1));
                        lnc.mean[a] = st.x;
                        lnc.rstd[a] = st.y;
                    }
                }
                if constexpr (LN == LN_HEAD) {         // the head's weight fragments in the packed accumulators' k order (epilogue_head)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const unsigned short* hp = (const unsigned short*)d.head_w + (long)l15 * BN + (n0 & (BN - 1)) + 32 * t + grp * 4;
                        const uint2 lo = *(const uint2*)hp, hi = *(const uint2*)(hp + 16);
                        headw[t] = u32x4{lo.x, lo.y, hi.x, hi.y};
                    }
                }
            }
            mfma_quadrant(1, 1);
            // phase 4: quadrant (1,0)
            read_w(0);
            issue(P_W1);
            // vmcnt(4): all but the two newest pieces have landed.  Tile's last slab: vmcnt(2) -- the bias vector was requested
            // between those two pieces
            if (kt == nk - 1) __builtin_amdgcn_s_waitcnt(0x0F72);
            else __builtin_amdgcn_s_waitcnt(0x0F74);
            asm volatile("" ::: "memory");
            if (kt == 0) MAGE_STAMP(it, 2);             // probe: the tile's first counted wait has completed
            mfma_quadrant(1, 0);
            if (kt == 0) MAGE_STAMP(it, 3);             // probe: end of the tile's first slab
            c_buf ^= 1;
        }
        if (!wr) ring_barrier();                       // the leading half waits for the trailing half's last MFMA section
#if MAGE_ABL == 1
        {   // tuning build: main loop only (keep the accumulators alive, store nothing)
            float sacc = 0.f;
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) sacc += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
            if (sacc == 123456.789f) ((float*)d.Y)[0] = sacc + biasm[0][0];
            continue;
        }
#endif
        MAGE_STAMP(it, 0);                             // probe: K loop done
        char* stg = smem + 2 * KBUF + wave * 4096;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));               // keep the epilogue's lane-derived constants out of the K loop's registers
        if constexpr (LN == LN_HEAD) {
            epilogue_head<ACT>(d, biasm, acc, headw, tm * BM, lane_e, wave, plane, smem + 2 * KBUF,
                               d.head_phases ? (long)(tn >> 1) * (d.y_mul_y >> 1) + (long)(tn & 1) * (d.y_mul_x >> 1) : 0L);
        } else if constexpr (LN == LN_CONSUME) {
            if (d.y_dtype == MAGE_F32) epilogue_lean<ACT, float, MT, TAPS, LN>(d, biasm, acc, m0, n0, lane_e, plane, stg, ysplit, &lnc);
            else epilogue_lean<ACT, H16, MT, TAPS, LN>(d, biasm, acc, m0, n0, lane_e, plane, stg, ysplit, &lnc);
        } else if constexpr (LN == LN_PRODUCE) {
            // fp32 stream + 16-bit copy, or (16-bit y_dtype; always with RB: host check) the 16-bit stream alone
            if (RB || d.y_dtype != MAGE_F32) epilogue_lean<ACT, H16, MT, TAPS, LN, 0, H16, RB && !TAPS>(d, biasm, acc, m0, n0, lane_e, plane, stg, ysplit);
            else epilogue_lean<ACT, float, MT, TAPS, LN, 0, H16>(d, biasm, acc, m0, n0, lane_e, plane, stg, ysplit);
        } else if constexpr (LN == LN_DUAL || LN == LN_GELUBWD) {
            epilogue_lean<ACT, unsigned short, MT, TAPS, LN>(d, biasm, acc, m0, n0, lane_e, plane, stg, ysplit);               // bf16 rows (host check)
        } else if constexpr (SPL != 0) {
            if (d.y_dtype == MAGE_F32) epilogue_lean<ACT, float, MT, TAPS>(d, biasm, acc, m0, n0, lane_e, plane, stg, ysplit);
            else epilogue_lean<ACT, float, MT, TAPS, LN_NONE, SPL>(d, biasm, acc, m0, n0, lane_e, plane, stg, ysplit);   // split rows out
        } else {
            if (d.y_dtype == MAGE_F32) epilogue_lean<ACT, float, MT, TAPS>(d, biasm, acc, m0, n0, lane_e, plane, stg, ysplit);
            else epilogue_lean<ACT, H16, MT, TAPS, LN_NONE, 0, H16, RB && (!TAPS || EK == EK_BIAS)>(d, biasm, acc, m0, n0, lane_e, plane, stg, ysplit);
        }
        MAGE_STAMP(it, 1);                             // probe: epilogue issued
    }
}


// ======================================================================================================================
// gemm_small_kernel: the bf16 plain GEMM when there are too few rows for the tiled kernels to fill the chip (the reference samples
// ONE clip per call, main_mage.py:205,239-241: an incremental step is M = 256 rows, i.e. 2-16 tiles of the kernels above on 256 CUs,
// each walking its K slabs at the LDS-DMA round trip: 13-32 µs per launch).  Here the output is cut into 16*RW x 64 pieces, one
// 4-wave workgroup each; wave w owns the 16 columns nt = w of the piece for all RW row blocks and streams its operands STRAIGHT from
// global memory / L2 into the MFMA operand layout (lane (l15, grp) holds 8 consecutive k of row / column l15: one 16-byte load per
// fragment; the four waves' A loads hit the same lines), two rounds of U k-steps in flight in registers, no LDS and no barrier in
// the K loop.  Then waves 1-3 hand their accumulators to wave 0 through LDS and wave 0 runs epilogue_lean -- the SAME function on the
// SAME accumulator layout as the tiled kernels -- so with the same MFMA and the same k order (k-steps of 32 ascending) every output
// element gets the same bits from all three kernels: B = 1 == row 0 of a batch, incremental == full loop stay exact.
// SPL = 2: split-precision operands of f16 pieces (the fast parity mode at one clip per call): the k-steps walk the physical
// [hi(64) | lo(64)] slabs in the tiled kernels' order -- (A_hi, W_lo), (A_lo, W_hi) per logical slab, accumulators * 2^-11, then the
// (A_hi, W_hi) slabs -- on v_mfma_f32_16x16x32_f16.
template <int ACT, int EK, int LN, bool RB, int RW, int SPL = 0, bool HF = false>
__global__ __launch_bounds__(256) void gemm_small_kernel(const mage_gemm_desc d) {
    static_assert(EK != EK_GENERAL && (!RB || EK == EK_RES_INIT), "lean epilogue kinds");
    static_assert(!HF || SPL == 0, "f16 form: plain operands");
    typedef H16T<HF> H16;
    static_assert(SPL == 0 || (SPL == 2 && LN == LN_NONE && !RB), "split-precision form: f16 pieces, no LayerNorm fold");
    constexpr int D = RW == 1 ? 16 : RW == 2 ? 8 : 4;  // k-steps (32 columns) in flight; (K / 32) % D == 0 (host check)
    __shared__ __attribute__((aligned(16))) char sm[4096 + 3 * RW * 1024];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, grp = lane >> 4;
    const int ntn = (d.N + 63) >> 6;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
    const int m0 = tm * 16 * RW, n0 = tn * 64;
    const int plane = d.out_h * d.out_w;
    const unsigned short* ap[RW];
#pragma unroll
    for (int a = 0; a < RW; ++a) {
        const int m = min(m0 + a * 16 + l15, d.M - 1);
        const int img = m / plane, rem = m - img * plane;
        const int oy = rem / d.out_w, ox = rem - oy * d.out_w;
        ap[a] = (const unsigned short*)d.A + ((long)img * d.a_img_stride + (long)oy * d.in_w + ox + d.a_off) * d.lda + grp * 8;
    }
    const unsigned short* wp = (const unsigned short*)d.W + (long)min(n0 + w * 16 + l15, d.N - 1) * (SPL ? d.ldw : d.K) + grp * 8;
    [[maybe_unused]] const int nk2s = SPL ? (d.K >> 6) * 4 : 0;      // SPL: k-steps of the two small-term passes
    f32x4 acc[RW];
    constexpr bool res_epi = RB;                       // 16-bit stream, 16-bit rows out (host check): residual added in the epilogue (RESE)
    if constexpr (EK == EK_RES_INIT && res_epi) {
#pragma unroll
        for (int a = 0; a < RW; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if constexpr (EK == EK_RES_INIT) {
        const int nn = n0 + w * 16 + grp * 4;
#pragma unroll
        for (int a = 0; a < RW; ++a) {
            const long row = (long)min(m0 + a * 16 + l15, d.M - 1) * d.y_mul_x + d.y_off;
            acc[a] = *(const f32x4*)((const float*)d.residual + row * d.ldr + (nn < d.N ? nn : 0));
        }
        if constexpr (SPL == 2) {                      // the residual gets the lo pieces' scale (exact), undone at k-step nk2s
#pragma unroll
            for (int a = 0; a < RW; ++a) acc[a] *= MAGE_F16_LO_SCALE;
        }
    } else {
#pragma unroll
        for (int a = 0; a < RW; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the epilogue's vectors, requested first (they are one more round trip when asked for after the K loop)
    f32x4 biasm[4];
    [[maybe_unused]] LnConsume lnc;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int n = n0 + b * 16 + grp * 4;
        biasm[b] = d.bias ? *(const f32x4*)(d.bias + (n < d.N ? n : 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (LN == LN_CONSUME) lnc.s[b] = *(const f32x4*)(d.ln_colsum + (n < d.N ? n : 0));
    }
    if constexpr (LN == LN_CONSUME) {
        if (d.ln_stats) {
#pragma unroll
            for (int a = 0; a < RW; ++a) {
                const float2 st = *(const float2*)(d.ln_stats + 2 * (long)min(m0 + a * 16 + l15, d.M - 1));
                lnc.mean[a] = st.x;
                lnc.rstd[a] = st.y;
            }
        } else if (w == 0) {                           // straight from the producer's partial sums (mage_ln_stats' arithmetic)
            const int ns = d.K >> 6;
#pragma unroll
            for (int a = 0; a < RW; ++a)
                mage_ln_stats_row((const float2*)d.ln_part + (long)min(m0 + a * 16 + l15, d.M - 1), d.ln_part_rows, ns, 1.0f / (float)d.K, d.ln_eps,
                                  lnc.mean[a], lnc.rstd[a]);
        }
    }
    // a ring of D k-steps in registers: slot u is multiplied and at once re-requested D steps ahead (the loads return in order, so
    // hipcc's counted vmcnt lets each MFMA start as soon as its own operands are there)
    uint4 xa[D][RW], wb[D];
    auto request = [&](int u, int ks) __attribute__((always_inline)) {
        int ao = ks * 32, wo = ks * 32;
        if constexpr (SPL != 0) {                      // physical 64-element unit along the split row (gemm_kernel's issue_one)
            const int kt = ks >> 1, t32 = (ks & 1) * 32;
            const int ua = ks < nk2s ? kt : 2 * (kt - (nk2s >> 1));
            const int uw = ks < nk2s ? (kt ^ 1) : ua;
            ao = ua * 64 + t32;
            wo = uw * 64 + t32;
        }
#pragma unroll
        for (int a = 0; a < RW; ++a) xa[u][a] = *(const uint4*)(ap[a] + ao);
        wb[u] = *(const uint4*)(wp + wo);
    };
    auto multiply = [&](int u, int ks) __attribute__((always_inline)) {
        if constexpr (SPL == 2) {
            if (ks == nk2s) {
#pragma unroll
                for (int a = 0; a < RW; ++a) acc[a] *= (1.0f / MAGE_F16_LO_SCALE);
            }
        }
#pragma unroll
        for (int a = 0; a < RW; ++a) {
            if constexpr (SPL == 2)
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wb[u]), __builtin_bit_cast(f16x8, xa[u][a]), acc[a], 0, 0, 0);
            else
                acc[a] = mfma16x16x32<H16>(__builtin_bit_cast(u32x4, wb[u]), __builtin_bit_cast(u32x4, xa[u][a]), acc[a]);
        }
    };
    const int nks = SPL ? (d.K >> 6) * 6 : d.K >> 5;
#pragma unroll
    for (int u = 0; u < D; ++u) request(u, u);
    int ks0 = 0;
    for (; ks0 + D < nks; ks0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            multiply(u, ks0 + u);
            request(u, ks0 + D + u);
        }
    }
#pragma unroll
    for (int u = 0; u < D; ++u) multiply(u, ks0 + u);
    // waves 1..3 -> wave 0
    if (w) {
#pragma unroll
        for (int a = 0; a < RW; ++a) *(f32x4*)(sm + 4096 + ((w - 1) * RW + a) * 1024 + lane * 16) = acc[a];
    }
    __syncthreads();
    if (w) return;
    f32x4 accf[RW][4];
#pragma unroll
    for (int a = 0; a < RW; ++a) {
        accf[a][0] = acc[a];
#pragma unroll
        for (int nt = 1; nt < 4; ++nt) accf[a][nt] = *(const f32x4*)(sm + 4096 + ((nt - 1) * RW + a) * 1024 + lane * 16);
    }
    if constexpr (SPL != 0) {
        if (d.y_dtype == MAGE_F32) epilogue_lean<ACT, float, RW>(d, biasm, accf, m0, n0, lane, plane, sm, 0);
        else epilogue_lean<ACT, float, RW, false, LN_NONE, SPL>(d, biasm, accf, m0, n0, lane, plane, sm, 0);     // split rows out
    } else if constexpr (LN == LN_CONSUME) {
        if (d.y_dtype == MAGE_F32) epilogue_lean<ACT, float, RW, false, LN>(d, biasm, accf, m0, n0, lane, plane, sm, 0, &lnc);
        else epilogue_lean<ACT, H16, RW, false, LN>(d, biasm, accf, m0, n0, lane, plane, sm, 0, &lnc);
    } else {
        if (d.y_dtype == MAGE_F32) epilogue_lean<ACT, float, RW, false, LN, 0, H16>(d, biasm, accf, m0, n0, lane, plane, sm, 0);
        else epilogue_lean<ACT, H16, RW, false, LN, 0, H16, RB>(d, biasm, accf, m0, n0, lane, plane, sm, 0);
    }
}

template <int ACT, int EK, int LN, bool RB, int SPL = 0, bool HF = false>
int launch_small(const mage_gemm_desc* d, hipStream_t s, int n_cu) {
    const long p64 = (long)((d->N + 63) / 64);
    // rows per workgroup: as many as still leave ~2 workgroups per CU (fewer re-reads of W from L2)
    const int rw = ((long)((d->M + 63) / 64) * p64 >= 2L * n_cu) ? 4 : ((long)((d->M + 31) / 32) * p64 >= 2L * n_cu) ? 2 : 1;
    const unsigned grid = (unsigned)(((d->M + 16 * rw - 1) / (16 * rw)) * p64);
    if (rw == 4) hipLaunchKernelGGL((gemm_small_kernel<ACT, EK, LN, RB, 4, SPL, HF>), dim3(grid), dim3(256), 0, s, *d);
    else if (rw == 2) hipLaunchKernelGGL((gemm_small_kernel<ACT, EK, LN, RB, 2, SPL, HF>), dim3(grid), dim3(256), 0, s, *d);
    else hipLaunchKernelGGL((gemm_small_kernel<ACT, EK, LN, RB, 1, SPL, HF>), dim3(grid), dim3(256), 0, s, *d);
    MAGE_CHECK_LAUNCH("mage_gemm");
    return MAGE_OK;
}

template <int DT, bool GATHER, int ACT, int MT, int EK, bool SPLIT = false, int LN = LN_NONE, int NW = 4, int SPL = 0, bool RB = false>
int launch_tile(const mage_gemm_desc* d, hipStream_t s, int n_cu) {
    typedef Tile<MT, NW, ring_stages<DT, GATHER, MT, EK, SPLIT, NW>()> TL;
    // launch attributes are per DEVICE (a process may drive several GPUs, e.g. nn.DataParallel, main_mage.py:106): cached per
    // device index; setting one twice from two threads is harmless
    static bool attr_set[MAGE_MAX_DEVICES] = {false};
    const int dev = mage_device_index();
    MAGE_CHECK_ARG(dev >= 0, "mage_gemm: no current device");
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<DT, GATHER, ACT, MT, EK, SPLIT, LN, NW, SPL, RB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  TL::LDS_BYTES);
        attr_set[dev] = true;
    }
    GemmArgs a;
    a.d = *d;
    a.zero = (const char*)mage_zero_page();
    const int tiles_m = (d->M + TL::BM - 1) / TL::BM;
    a.ntiles_n = (d->N + TL::BNT - 1) / TL::BNT;
    a.tiles_per_split = tiles_m * a.ntiles_n;
    a.ntiles = a.tiles_per_split * d->n_split;
    const int grid = a.ntiles >= n_cu ? n_cu : ((a.ntiles + 7) & ~7);     // one resident workgroup per CU, multiple of 8
    // staggered start (see gemm_kernel): G groups spread over a fraction of one estimated tile period = K loop (~3.4 k clocks
    // per 64-wide slab of a 256x256 tile, measured) + the tile's HBM burst at the all-at-once rate (~10.6 B per clock per CU,
    // measured).  Only when every workgroup has enough tiles for the idle start to pay.  MAGE_GEMM_STAGGER="G,percent"
    // overrides (tuning), "0" disables.  Measured on the decoder's 4-GEMM block: 775 -> 800 TFLOP/s with 8 groups over 60 %.
    const int st_groups = mage_options().gemm_stagger_groups, st_percent = mage_options().gemm_stagger_percent, st_env = mage_options().gemm_stagger_forced;
    a.stagger_groups = 0;
    a.stagger_sleeps = 0;
    const int tiles_per_wg = a.ntiles / grid;
    // Only the residual kind by default: its tile ends in a 512 KB read + write burst per CU that the stagger spreads (out_proj
    // 0.307 -> 0.276 ms, c_proj 0.567 -> 0.542 ms).  The bias kinds (QKV, c_fc) have no drain stall to hide (round-2 tile probe); an
    // interleaved A/B with and without it is inside +-0.5 %, so they skip the idle start.  MAGE_GEMM_STAGGER applies to every kind.
    // Round 6 (tools/producer_probe.py, profiles/r06_producer_stagger.txt, 8 interleaved rounds): out_proj (K = 512) 189.6 us with the default
    // 8 groups over 60 % against 197.2 without; c_proj (K = 2048: the burst is a quarter of the tile period) 500.3 with it, 484.9 WITHOUT: the
    // idle start only pays for short K loops, so it is applied up to 16 slabs (K <= 1024 16-bit elements).
    const int es = d->dtype == MAGE_F32 ? 4 : 2;
    const long nk = ((long)d->K * es + 127) / 128 * (SPL ? 3 : 1);
    const bool kind_wants = st_env || (EK == EK_RES_INIT && nk <= 16);
    if (kind_wants && st_groups > 1 && a.ntiles >= n_cu && tiles_per_wg >= 6) {
        const long out_b = (long)TL::BM * TL::BNT * (d->y_dtype == MAGE_F32 ? 4 : 2);
        const long res_b = d->residual ? (long)TL::BM * TL::BNT * (d->res_dtype == MAGE_F32 ? 4 : 2) : 0;
        const long period = nk * 3400 * MT / 8 + (long)((out_b + res_b) / 10.6);
        a.stagger_groups = st_groups;
        a.stagger_sleeps = (int)(period * st_percent / 100 / st_groups / 1024);
    }
    if constexpr (DT != MAGE_F32 && !GATHER && MT == 8 && EK != EK_GENERAL && NW == 4) {
        constexpr bool HF = DT == MAGE_F16;
        const int use8 = !mage_options().gemm_no_8phase;
        const long a_rows = (long)((d->M + d->out_h * d->out_w - 1) / (d->out_h * d->out_w)) * d->a_img_stride + d->a_off + 1;
        const long a_span = a_rows * d->lda + (long)(d->n_split - 1) * d->a_split_stride;       // elements reachable from A / W
        const long w_span = (long)d->N * d->ldw + (long)(d->n_split - 1) * d->w_split_stride;
        if (use8 && d->K % 64 == 0 && a_span * 2 < (1L << 32) && w_span * 2 < (1L << 32)) {
            static bool attr8[MAGE_MAX_DEVICES] = {false};
            if (!attr8[dev]) {
                (void)hipFuncSetAttribute((const void*)gemm8_kernel<ACT, EK, SPLIT, false, LN, SPL, RB, HF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024);
                attr8[dev] = true;
            }
            hipLaunchKernelGGL((gemm8_kernel<ACT, EK, SPLIT, false, LN, SPL, RB, HF>), dim3(grid), dim3(512), 160 * 1024, s, a);
            MAGE_CHECK_LAUNCH("mage_gemm");
            return MAGE_OK;
        }
    }
    hipLaunchKernelGGL((gemm_kernel<DT, GATHER, ACT, MT, EK, SPLIT, LN, NW, SPL, RB>), dim3(grid), dim3(512), TL::LDS_BYTES, s, a);
    MAGE_CHECK_LAUNCH("mage_gemm");
    return MAGE_OK;
}


// Padded-taps convolutions on the 8-phase kernel (gemm8_kernel TAPS): eligible shapes only; returns 1 if launched, 0 if not
// eligible (the caller falls through to the generic gather kernel), < 0 on error.
template <int ACT, int EK, int SPL = 0, int LN = LN_NONE, bool HF = false, bool RB = false>
int launch_taps8(const mage_gemm_desc* d, hipStream_t s, int n_cu) {
    const int dev = mage_device_index();
    MAGE_CHECK_ARG(dev >= 0, "mage_gemm: no current device");
    static bool attr[MAGE_MAX_DEVICES] = {false};
    if (!attr[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm8_kernel<ACT, EK, false, true, LN, SPL, RB, HF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr[dev] = true;
    }
    GemmArgs a;
    a.d = *d;
    a.zero = (const char*)mage_zero_page();
    const int tiles_m = (d->M + 255) / 256;
    a.ntiles_n = (d->N + BN - 1) / BN;
    a.tiles_per_split = tiles_m * a.ntiles_n;
    a.ntiles = a.tiles_per_split;
    a.stagger_groups = 0;
    a.stagger_sleeps = 0;
    const int grid = a.ntiles >= n_cu ? n_cu : ((a.ntiles + 7) & ~7);
    hipLaunchKernelGGL((gemm8_kernel<ACT, EK, false, true, LN, SPL, RB, HF>), dim3(grid), dim3(512), 160 * 1024, s, a);
    MAGE_CHECK_LAUNCH("mage_gemm");
    return 1;
}

template <int SPL = 0, bool HF = false>
int try_taps8(const mage_gemm_desc* d, hipStream_t s) {
    const int use8 = !(mage_options().gemm_no_8phase || mage_options().gemm_no_taps8);
    if ((!use8 && SPL == 0) || d->dtype != (SPL == 0 ? (HF ? MAGE_F16 : MAGE_BF16) : SPL == 1 ? MAGE_BF16X3 : MAGE_F16X3) || d->n_split != 1) return 0;
    const bool table = d->rowadd && !d->residual;                                                         // y = table[row] + conv (+ bias)
    const bool plain = !d->rowadd && !d->residual;
    const int ntaps = d->taps_h * d->taps_w;
    // convolutions over a zero-padded input, and (taps 1x1) the Linear layers that add a broadcast row table (in_linear /
    // context_linear + T positions, written into regrouped rows of the decoder stream)
    if (ntaps <= 1 && !table) return 0;
    if (d->stride != 1 || d->dys != 1 || d->dxs != 1 || d->dy0 != 0 || d->dx0 != 0 || d->a_half) return 0;
    // zero-padded input only: every tap of every output pixel is a row of the buffer (in_w is the buffer's row pitch: a window that
    // starts inside the padding -- the sub-pixel phases of a transposed convolution -- comes with a_off and a wider pitch)
    if (d->in_h < d->out_h + d->taps_h - 1 || d->in_w < d->out_w + d->taps_w - 1) return 0;
    if (d->cin % 64 != 0 || d->K % 64 != 0 || d->scale || d->post_relu) return 0;
    int dev = mage_device_index();
    if (dev < 0) return 0;
    hipDeviceProp_t p;
    static int n_cu_dev[MAGE_MAX_DEVICES] = {0};
    if (!n_cu_dev[dev]) n_cu_dev[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount >= 8) ? (p.multiProcessorCount & ~7) : 256;
    const int n_cu = n_cu_dev[dev];
    // any tile count (also the 64-image conv of the incremental AR mode: both modes must run the SAME arithmetic, their tokens
    // are compared bitwise), but only the widths the 8-phase kernel is exercised at
    if (d->N % 256 != 0 || d->M % 256 != 0) return 0;
    const long n_img = (d->M + (long)d->out_h * d->out_w - 1) / ((long)d->out_h * d->out_w);
    if (d->a_img_stride < (long)(d->out_h + d->taps_h - 2) * d->in_w + d->out_w + d->taps_w - 1) return 0;   // the caller's padded image
    const long a_span = (n_img * d->a_img_stride + d->a_off + (long)d->in_h * d->in_w) * d->lda;
    if (a_span * 2 >= (1L << 32) || (long)d->N * (SPL ? d->ldw : d->K) * 2 >= (1L << 32)) return 0;
    if (d->head_w) {
        // the narrow Linear on the tile's rows (LN_HEAD): one column tile must hold whole rows; refused loudly, the caller asked for a fusion
        MAGE_CHECK_ARG(d->head_phases == 0 || (d->head_phases == 4 && d->taps_h == 2 && d->taps_w == 2 && d->y_mul_x == 2 && d->y_mul_y % 2 == 0),
                       "mage_gemm: head_phases is 0 or 4 (the sub-pixel phases of a 4 x 4 / stride 2 transposed convolution: 2 x 2 taps, y_mul_x = 2)");
        const bool head_res = !d->rowadd && d->residual && d->res_dtype == MAGE_BF16 && d->head_phases == 0 && d->ldr % 4 == 0 && d->ldr >= d->N &&
                              (((uintptr_t)d->residual) & 7) == 0 && (d->res_half || (d->out_h * d->out_w == d->y_img_stride && d->y_off == 0));
        MAGE_CHECK_ARG(SPL == 0 && (plain || head_res) && d->act == MAGE_ACT_RELU && d->N == (d->head_phases ? 1024 : 256) && d->y_dtype == MAGE_F32 && (d->ldy >= 16 || (d->ldy == 4 && d->head_phases == 0)) &&
                           d->ldy % 4 == 0 && d->bias && (((uintptr_t)d->head_w) & 7) == 0 && (((uintptr_t)d->Y) & 15) == 0,
                       "mage_gemm: head_w takes the bf16 padded-taps form with N = 256 (1024 with head_phases), bias, ReLU, fp32 [row][ldy >= 16] output (ldy == 4: taps 0..3 only), "
                       "optionally a bf16 residual (N = 256 form; res_half or packed rows)");
        if constexpr (SPL == 0 && !HF) return launch_taps8<MAGE_ACT_RELU, EK_BIAS, 0, LN_HEAD>(d, s, n_cu);
    }
    if (table && d->act == MAGE_ACT_NONE && (((uintptr_t)d->rowadd) & 15) == 0) return launch_taps8<MAGE_ACT_NONE, EK_RES_INIT, SPL, LN_NONE, HF>(d, s, n_cu);
    if constexpr (SPL == 0 && !HF) {
        // a convolution that adds a bf16 residual tensor of its own (optionally at half resolution) and writes bf16 rows: a bottleneck block's
        // closing convolution + identity path (vqvae_model.py:147-166); the residual rows are fetched in the epilogue (epilogue_lean RESE)
        if (!d->rowadd && d->residual && d->res_dtype == MAGE_BF16 && d->y_dtype == MAGE_BF16 && d->act == MAGE_ACT_NONE && d->bias && d->ldr % 8 == 0 &&
            d->ldr >= d->N && (((uintptr_t)d->residual) & 15) == 0 && d->y_mul_x == 1 &&
            // (without res_half the residual is indexed like the OUTPUT rows, as in the general epilogue: here only where that is the plain row m)
            (d->res_half || (d->out_h * d->out_w == d->y_img_stride && d->y_mul_y == d->out_w && d->y_off == 0)))
            return launch_taps8<MAGE_ACT_NONE, EK_BIAS, 0, LN_NONE, false, true>(d, s, n_cu);
    }
    if constexpr (HF) return 0;                        // f16: the row-table forms only (context_linear / in_linear / the frame convolution + positions)
    if constexpr (SPL == 0) {
        if (plain && d->act == MAGE_ACT_NONE) return launch_taps8<MAGE_ACT_NONE, EK_BIAS>(d, s, n_cu);
        if (plain && d->act == MAGE_ACT_RELU) return launch_taps8<MAGE_ACT_RELU, EK_BIAS>(d, s, n_cu);
    }
    if constexpr (SPL == 2) {      // f16 pieces: the VQ-VAE encoder's convolutions (bias with BatchNorm folded in, ReLU)
        if (plain && d->act == MAGE_ACT_NONE) return launch_taps8<MAGE_ACT_NONE, EK_BIAS, 2>(d, s, n_cu);
        if (plain && d->act == MAGE_ACT_RELU) return launch_taps8<MAGE_ACT_RELU, EK_BIAS, 2>(d, s, n_cu);
    }
    return 0;
}

// few rows: the 128-row tile list would cover less than half of the chip (one clip per call; see gemm_small_kernel)
bool small_shape(int M, int N, int K, int n_cu) {
    const int small = !mage_options().gemm_no_small, small_m = mage_options().gemm_small_m;
    const long tiles4 = (long)((M + 127) / 128) * ((N + BN - 1) / BN);
    return small && 2 * tiles4 <= n_cu && K % 512 == 0 && N % 16 == 0 && M <= small_m;
}

// Split-precision GEMMs (dtype MAGE_BF16X3 / MAGE_F16X3): the decoder's Linear layers and frame convolution in the fast parity mode.
template <int SPL>
int launch_spl(const mage_gemm_desc* d, hipStream_t s) {
    static int n_cu_dev[MAGE_MAX_DEVICES] = {0};
    const int dev = mage_device_index();
    MAGE_CHECK_ARG(dev >= 0, "mage_gemm: no current device");
    if (!n_cu_dev[dev]) {
        hipDeviceProp_t p;
        int n = 256;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount >= 8) n = p.multiProcessorCount & ~7;
        n_cu_dev[dev] = n;
    }
    const int n_cu = n_cu_dev[dev];
    MAGE_CHECK_ARG(!d->scale && !d->post_relu && !d->y2 && !d->ln_stats && d->n_split == 1 && !d->a_half && !d->res_half,
                   "mage_gemm: split-precision form: epilogue y = act(acc + bias) | residual + acc + bias | rowadd[..] + acc only");
    MAGE_CHECK_ARG(!d->rowadd, "mage_gemm: split-precision form: row tables only in the padded-taps form");
    const bool big = (long)((d->M + 255) / 256) * ((d->N + BN - 1) / BN) >= 2L * n_cu;
    // few rows (one clip per call): gemm_small_kernel, f16 pieces
    bool few = false;
    if constexpr (SPL == 2) few = small_shape(d->M, d->N, d->K, n_cu);
    if (d->residual) {
        MAGE_CHECK_ARG(d->res_dtype == MAGE_F32 && d->act == MAGE_ACT_NONE && d->out_h == 1 && d->out_w >= d->M && (((uintptr_t)d->residual) & 15) == 0,
                       "mage_gemm: split-precision form: the residual is the fp32 stream (plain rows, no activation)");
        if constexpr (SPL == 2) {
            if (few) return launch_small<MAGE_ACT_NONE, EK_RES_INIT, LN_NONE, false, 2>(d, s, n_cu);
        }
        return big ? launch_tile<MAGE_BF16, false, MAGE_ACT_NONE, 8, EK_RES_INIT, false, LN_NONE, 4, SPL>(d, s, n_cu)
                   : launch_tile<MAGE_BF16, false, MAGE_ACT_NONE, 4, EK_RES_INIT, false, LN_NONE, 4, SPL>(d, s, n_cu);
    }
    if constexpr (SPL == 2) {
        if (few && d->act == MAGE_ACT_NONE) return launch_small<MAGE_ACT_NONE, EK_BIAS, LN_NONE, false, 2>(d, s, n_cu);
        if (few && d->act == MAGE_ACT_QUICKGELU) return launch_small<MAGE_ACT_QUICKGELU, EK_BIAS, LN_NONE, false, 2>(d, s, n_cu);
    }
    if (d->act == MAGE_ACT_NONE)
        return big ? launch_tile<MAGE_BF16, false, MAGE_ACT_NONE, 8, EK_BIAS, false, LN_NONE, 4, SPL>(d, s, n_cu)
                   : launch_tile<MAGE_BF16, false, MAGE_ACT_NONE, 4, EK_BIAS, false, LN_NONE, 4, SPL>(d, s, n_cu);
    if (d->act == MAGE_ACT_QUICKGELU)
        return big ? launch_tile<MAGE_BF16, false, MAGE_ACT_QUICKGELU, 8, EK_BIAS, false, LN_NONE, 4, SPL>(d, s, n_cu)
                   : launch_tile<MAGE_BF16, false, MAGE_ACT_QUICKGELU, 4, EK_BIAS, false, LN_NONE, 4, SPL>(d, s, n_cu);
    mage_set_error("mage_gemm: split-precision form: activation %d is not available (none / QuickGELU)", d->act);
    return MAGE_EINVAL;
}

template <int DT, bool GATHER, int ACT, int EK, int LN = LN_NONE, bool RB = false>
int launch_ek(const mage_gemm_desc* d, hipStream_t s) {
    static int n_cu_dev[MAGE_MAX_DEVICES] = {0};
    const int dev = mage_device_index();
    MAGE_CHECK_ARG(dev >= 0, "mage_gemm: no current device");
    if (!n_cu_dev[dev]) {
        hipDeviceProp_t p;
        int n = 256;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount >= 8) n = p.multiProcessorCount & ~7;
        n_cu_dev[dev] = n;
    }
    const int n_cu = n_cu_dev[dev];
    if constexpr (DT != MAGE_F32 && !GATHER && EK != EK_GENERAL && LN != LN_DUAL && LN != LN_GELUBWD && (ACT == MAGE_ACT_NONE || ACT == MAGE_ACT_QUICKGELU)) {
        if (d->n_split == 1 && !d->a_relu && small_shape(d->M, d->N, d->K, n_cu)) return launch_small<ACT, EK, LN, RB, 0, DT == MAGE_F16>(d, s, n_cu);
    }
    // 256-row tiles only where there are enough of them to give every CU at least two (bf16; the fp32 8x4-accumulator
    // variant does not fit the register file)
    if constexpr (EK != EK_RES_INIT && LN == LN_NONE) {
        // narrow outputs (N <= 128) on the 256 x 64 tile: a 256-column tile would spend 3/4 (N = 64) or more of its matrix-core work on
        // columns that do not exist.  Only where a 256-row tile list still fills the chip.
        const int narrow = !mage_options().gemm_no_narrow;
        if (narrow && d->N <= 128 && d->n_split == 1 && (long)((d->M + 255) / 256) * ((d->N + 63) / 64) >= n_cu)
            return launch_tile<DT, GATHER, ACT, 2, EK, false, LN_NONE, 1>(d, s, n_cu);
    }
    // (a_relu is applied to the operand fragments of that tile's K loop only: refused, not dropped, when the shape does not run there)
    MAGE_CHECK_ARG(!d->a_relu, "mage_gemm: a_relu runs on the 256 x 64 tile: N <= 128, at least one tile per CU, option gemm_no_narrow off");
    if constexpr (DT != MAGE_F32 && !GATHER && ACT == MAGE_ACT_NONE && EK == EK_RES_INIT) {
        // few rows (the incremental AR loop's x + Linear(.) at 8 k rows x 512 columns: 128 tiles of 128 x 256 on 256 CUs): the narrow tile
        // cuts the same output into 256 x 64 pieces, one per CU.  Same K order per element: the tokens stay bit-identical to the full loop's.
        const int few = !(mage_options().gemm_no_narrow || mage_options().gemm_no_narrow_few);
        const long tiles4 = (long)((d->M + 127) / 128) * ((d->N + BN - 1) / BN);
        // (tiles4 == n_cu, the B = 64 incremental step, measured on the narrow tile: 31.5 vs 28.2 ms per call -- the 128 x 256 tile stays)
        if (few && d->n_split == 1 && tiles4 < n_cu && d->N % 64 == 0) return launch_tile<DT, GATHER, ACT, 2, EK, false, LN, 1, 0, RB>(d, s, n_cu);
    }
    const long tiles256 = (long)((d->M + 255) / 256) * ((d->N + BN - 1) / BN) * d->n_split;
    if constexpr (!GATHER && ACT == MAGE_ACT_NONE && EK == EK_BIAS && DT != MAGE_F16) {
        if (d->n_split > 1) {                      // split-K (weight gradients): its own instantiations
            if constexpr (DT == MAGE_BF16) {
                if (tiles256 >= 2L * n_cu) return launch_tile<DT, GATHER, ACT, 8, EK, true>(d, s, n_cu);
            }
            return launch_tile<DT, GATHER, ACT, 4, EK, true>(d, s, n_cu);
        }
    }
    if constexpr (DT != MAGE_F32) {
        if (tiles256 >= 2L * n_cu) return launch_tile<DT, GATHER, ACT, 8, EK, false, LN, 4, 0, RB>(d, s, n_cu);
    }
    return launch_tile<DT, GATHER, ACT, 4, EK, false, LN, 4, 0, RB>(d, s, n_cu);
}

template <int DT, bool GATHER, int ACT>
int launch_act(const mage_gemm_desc* d, hipStream_t s) {
    const bool extras = d->scale || d->rowadd || d->residual || d->post_relu;
    if constexpr (DT != MAGE_F32 && !GATHER && (ACT == MAGE_ACT_NONE || ACT == MAGE_ACT_QUICKGELU)) {
        // LayerNorm folded around the GEMM (see epilogue_lean): whole interior tiles, rows not regrouped
        if constexpr (ACT == MAGE_ACT_QUICKGELU && DT == MAGE_BF16) {
            if (d->y2 && !d->ln_part && !d->ln_stats && !d->ln_colsum) {         // training: pre-activation rows + activated rows (LN_DUAL)
                MAGE_CHECK_ARG(d->M % 256 == 0 && d->N % 256 == 0 && d->n_split == 1 && !extras && d->out_h == 1 && d->out_w >= d->M && d->y_mul_x == 1 &&
                                   d->y_dtype == MAGE_BF16 && d->ldy2 % 8 == 0 && (((uintptr_t)d->y2) & 15) == 0,
                               "mage_gemm: y2 with QuickGELU (pre-activation + activated rows) needs bf16 plain rows, M and N multiples of 256");
                return launch_ek<DT, GATHER, ACT, EK_BIAS, LN_DUAL>(d, s);
            }
        }
        if (d->y2 || d->ln_stats || d->ln_part || d->ln_colsum) {
            MAGE_CHECK_ARG(d->M % 256 == 0 && d->N % 256 == 0 && d->n_split == 1,
                           "mage_gemm: the LayerNorm-folded forms need M and N multiples of 256");
            MAGE_CHECK_ARG(d->ln_stats || d->ln_colsum || (d->out_h == 1 && d->out_w >= d->M && d->y_mul_x == 1),
                           "mage_gemm: ln_part (LayerNorm partial sums of the new rows) needs plain output rows");
            if (d->ln_stats || d->ln_colsum) {
                MAGE_CHECK_ARG(!extras && d->ln_colsum && d->bias && !d->y2, "mage_gemm: ln_stats goes with ln_colsum and bias, nothing else");
                MAGE_CHECK_ARG(d->ln_stats || (d->ln_part && d->ln_eps > 0.f && mage_gemm_is_small(d->M, d->N, d->K) == 1),
                               "mage_gemm: a LayerNorm consumer without ln_stats needs ln_part + ln_eps and a few-rows size (mage_gemm_is_small)");
                return launch_ek<DT, GATHER, ACT, EK_BIAS, LN_CONSUME>(d, s);
            }
            if constexpr (ACT == MAGE_ACT_NONE) {
                // the producer forms of x + Linear(.): (fp32 residual) fp32 stream out + bf16 copy y2, or bf16 stream out alone (y2 null);
                // (bf16 residual) bf16 stream out
                MAGE_CHECK_ARG(d->residual && !d->scale && !d->rowadd && !d->post_relu && d->ln_part && (((uintptr_t)d->residual | (uintptr_t)d->y2) & 15) == 0,
                               "mage_gemm: ln_part goes with the residual form x + Linear(.)");
                MAGE_CHECK_ARG(d->y_dtype == MAGE_F32 ? (d->y2 && d->ldy2 % 8 == 0 && d->res_dtype == MAGE_F32) : !d->y2,
                               "mage_gemm: ln_part: fp32 stream out + 16-bit copy y2 (fp32 residual), or 16-bit stream out alone");
                if (d->res_dtype == DT) {
                    MAGE_CHECK_ARG(d->y_dtype == DT && d->ldr % 8 == 0, "mage_gemm: a 16-bit residual stream with ln_part writes 16-bit rows (ldr %% 8 == 0)");
                    return launch_ek<DT, GATHER, ACT, EK_RES_INIT, LN_PRODUCE, true>(d, s);
                }
                return launch_ek<DT, GATHER, ACT, EK_RES_INIT, LN_PRODUCE>(d, s);
            }
        }
    }
    MAGE_CHECK_ARG(!d->y2 && !d->ln_stats && !d->ln_part, "mage_gemm: y2 / ln_part / ln_stats are bf16 plain-GEMM options (act none or QuickGELU)");
    if (!extras) return launch_ek<DT, GATHER, ACT, EK_BIAS>(d, s);
    if constexpr (!GATHER && ACT == MAGE_ACT_NONE) {
        // the transformer's "x + Linear(.)": fp32 residual, nothing else after the bias, rows not regrouped
        if (d->residual && d->res_dtype == MAGE_F32 && !d->scale && !d->rowadd && !d->post_relu && d->out_h == 1 &&
            d->out_w >= d->M && (((uintptr_t)d->residual) & 15) == 0)
            return launch_ek<DT, GATHER, ACT, EK_RES_INIT>(d, s);
        if constexpr (DT != MAGE_F32) {            // the same on a 16-bit residual stream
            if (d->residual && d->res_dtype == DT && d->y_dtype == DT && !d->scale && !d->rowadd && !d->post_relu && d->out_h == 1 &&
                d->out_w >= d->M && (((uintptr_t)d->residual) & 15) == 0 && d->y_mul_x == 1 && d->ldr % 8 == 0)
                return launch_ek<DT, GATHER, ACT, EK_RES_INIT, LN_NONE, true>(d, s);
        }
    }
    return launch_ek<DT, GATHER, ACT, EK_GENERAL>(d, s);
}

template <int DT, bool GATHER>
int launch(const mage_gemm_desc* d, hipStream_t s) {
    if constexpr (DT == MAGE_F16) {                // f16: what the generation path uses (no ReLU / erf-GELU epilogues, gathers without activation)
        if (d->act == MAGE_ACT_NONE) return launch_act<DT, GATHER, MAGE_ACT_NONE>(d, s);
        if constexpr (!GATHER) {
            if (d->act == MAGE_ACT_QUICKGELU) return launch_act<DT, GATHER, MAGE_ACT_QUICKGELU>(d, s);
        }
        mage_set_error("mage_gemm: MAGE_F16 operands take act none (any geometry) or QuickGELU (plain rows), got act %d", d->act);
        return MAGE_EUNSUPPORTED;
    } else
    switch (d->act) {
        case MAGE_ACT_NONE: return launch_act<DT, GATHER, MAGE_ACT_NONE>(d, s);
        case MAGE_ACT_RELU: return launch_act<DT, GATHER, MAGE_ACT_RELU>(d, s);
        case MAGE_ACT_QUICKGELU: return launch_act<DT, GATHER, MAGE_ACT_QUICKGELU>(d, s);
        case MAGE_ACT_GELU_ERF: return launch_act<DT, GATHER, MAGE_ACT_GELU_ERF>(d, s);
        case MAGE_ACT_QUICKGELU_GRAD:
            if constexpr (DT == MAGE_BF16 && !GATHER) {
                MAGE_CHECK_ARG(d->y2 && d->y_dtype == MAGE_BF16 && d->M % 256 == 0 && d->N % 256 == 0 && d->n_split == 1 && !d->scale && !d->rowadd &&
                                   !d->residual && !d->post_relu && !d->ln_part && !d->ln_stats && !d->ln_colsum && d->out_h == 1 && d->out_w >= d->M &&
                                   d->y_mul_x == 1 && d->ldy2 % 8 == 0 && (((uintptr_t)d->y2) & 15) == 0,
                               "mage_gemm: MAGE_ACT_QUICKGELU_GRAD: y = acc * QuickGELU'(y2) on bf16 plain rows, M and N multiples of 256");
                return launch_ek<DT, GATHER, MAGE_ACT_NONE, EK_BIAS, LN_GELUBWD>(d, s);
            }
            mage_set_error("mage_gemm: MAGE_ACT_QUICKGELU_GRAD is a bf16 plain-GEMM epilogue");
            return MAGE_EINVAL;
        default: mage_set_error("mage_gemm: activation %d is not available in the GEMM epilogue", d->act); return MAGE_EINVAL;
    }
}

}  // namespace

