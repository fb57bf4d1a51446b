// Device helpers shared by the GEMM translation units (gemm.hip: the lockstep, 8-phase and few-rows kernels; gemm4.hip: the
// one-wave-per-SIMD kernel): LDS-DMA wrapper, activations, epilogue kinds, the lean epilogue, the bf16 residual helpers, the raw barrier.
#pragma once
#include <type_traits>
#include "common.h"

namespace {

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int ACT>
__device__ __forceinline__ float act_apply(float v) {
    if (ACT == MAGE_ACT_RELU) return fmaxf(v, 0.f);
    if (ACT == MAGE_ACT_QUICKGELU) return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669596f * v));
    if (ACT == MAGE_ACT_GELU_ERF) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    return v;
}

// EK (epilogue kind, compile time: every runtime "is this pointer set" test inside the row loop made hipcc fence the
// block with s_waitcnt vmcnt(0), i.e. wait for the previous row's store ack — 16 chained HBM round trips per tile):
//   EK_BIAS     y = act(acc + bias)                                   no loads in the epilogue at all
//   EK_RES_INIT same; the fp32 residual was loaded INTO the accumulators before the K loop (gemm_kernel)
//   EK_GENERAL  y = act((acc + bias)*scale + shift) + residual + rowadd   (VQ-VAE convolutions, positional tables)
enum { EK_BIAS = 0, EK_RES_INIT = 1, EK_GENERAL = 2 };

// Lean epilogue of the two kinds without loads: per (16-row tile mt, 32-column half k) 4 lane swaps, 4 packed bias adds, the
// activation, 4 packed converts and ONE 16-byte store (two for fp32 output) off a pointer that steps by 16 rows.  The
// per-wave cost of the general epilogue below was ~760 issued instructions (~4.5 k cycles, and the two waves of a SIMD run
// their epilogues back to back, not overlapped): 64-bit address arithmetic, the post-ReLU max and the row regrouping on
// every row.  Workgroup-edge tiles and regrouped rows take the predicated branch.
// ---- lean epilogue (EK_BIAS, EK_RES_INIT): y = act(acc + bias), no loads ------------------------------------------------
// What the stores look like to memory decides its cost.  The MFMA layout (and its permlane16_swap variant above) gives a
// wave-wide 16-byte store 16 rows x 64 bytes (bf16) or 64 scattered 16-byte pieces (fp32): 16-64 separate line accesses
// per instruction and half/eighth-filled 128-byte lines.  Measured in isolation (tools/probes/epi_probe.hip): ~275 cycles
// per store instruction, the two waves of a SIMD one after the other, 10.5 k cycles per 256x256 tile, 3.0 TB/s (bf16) and
// 2.4 TB/s (fp32) chip-wide; the same bytes as 8 rows x 128 contiguous bytes per instruction go at 22 cycles each, 5.2 TB/s.
// So each wave transposes its 16x64 block through a private 4 KiB LDS window (XOR-swizzled, conflict-free both ways; DS
// operations of one wave execute in order, so no barrier and no wait between its writes and reads) and stores whole rows:
//   bf16: one instruction = 8 rows x 128 B;   fp32: one instruction = 4 rows x 256 B.
// AFFINE (the TAPS instantiations of the 8-phase kernel): rows regrouped P at a time (out_h = 1, out_w = P, y_img_stride != P: the
// decoder stream's x[:, 1:] slots) still take the fast path when P % 256 == 0 -- a wave's 128 rows then lie in ONE group and
// yrow = m + group*(y_img_stride - P) + y_off.  A template parameter so that the Linear layers' kernels keep their exact code.
// LN: LayerNorm folded around the GEMMs (bf16 mode of the decoder stack; interior tiles and simple rows only -- the host checks).
//   LN_PRODUCE (x + Linear(.), fp32 stream out): the epilogue also writes a bf16 copy of the new x (y2) and, per row and per
//     64-column wave slice, the partial sums (sum x, sum x^2) of the fp32 values it holds anyway (ln_part[N/64][rows][2]):
//     mage_ln_stats turns them into (mean, rstd) per row.  The standalone LayerNorm pass (read 4 B + write 2 B per element) is gone.
//   LN_CONSUME (the Linear that follows the norm): A is that bf16 copy of x, W carries gamma (W' = gamma * W, rounded to bf16),
//     and the epilogue finishes the norm algebraically:   LN(x) W^T + b = rstd_m (x W'^T - mean_m s_n) + c_n,
//     s_n = sum_k W'_nk, c_n = sum_k beta_k W_nk + b_n (c arrives as `bias`).
struct LnConsume {
    float mean[8], rstd[8];            // per 16-row tile of the wave: the stats of this lane's row (row l15 of tile mt)
    f32x4 s[4];                        // s_n of this lane's 16 columns (MFMA layout)
};
// LN_DUAL (training, act = QuickGELU): TWO bf16 outputs from one accumulator tile -- y = the pre-activation rows (the backward pass needs
// them), y2 = QuickGELU(y) (the next Linear's operand): the forward activation pass (read 2 B + write 2 B per element) is gone.
// LN_GELUBWD (training, the data-gradient GEMM of c_proj): y = acc * QuickGELU'(aux), aux = the saved pre-activation rows (desc.y2, read
// here): the activation-backward pass (read 2 x 2 B + write 2 B per element) is gone; the product is taken on the fp32 accumulators.
// LN_HEAD (8-phase padded-taps kernel only, N = 256 = one column tile; desc.head_w): the rows y = act(acc + bias) are NOT stored.  The tile holds
// whole rows, so the narrow Linear that follows (16 outputs per row: the 4 x 4 taps of the VQ-VAE's last transposed convolution,
// vqvae_model.py:187) is taken on the bf16-rounded rows where they are: an MFMA whose B operand IS the packed accumulator layout (with a
// k-permuted weight fragment), a fixed-order sum over the four wave columns through LDS, fp32 [row][16] out (gemm.hip: epilogue_head).
enum { LN_NONE = 0, LN_PRODUCE = 1, LN_CONSUME = 2, LN_DUAL = 3, LN_GELUBWD = 4, LN_HEAD = 5 };

// OSPL (1 = bf16 pieces, 2 = f16 pieces; OT = float): the fp32 result leaves as a SPLIT row (common.h): a wave's 64 columns are one
// K slab of the consumer, [hi(64) | lo(64)] = the same 256 contiguous bytes per row as 64 fp32 values, so only the staging write differs.
// HT: the kernel's 16-bit element type (unsigned short = bf16, f16_t = f16): the type of the LN_PRODUCE copy y2 when OT is float, and of
// 16-bit rows OT themselves.
// RESE (16-bit rows out, LN_NONE / LN_PRODUCE; plain rows: host check): y = x + (A W^T + b) with the 16-bit residual rows x ADDED HERE instead
// of seeding the accumulators before the K loop (EK_RES_INIT).  The wave's residual block is requested in the epilogue itself, 16-row tile by
// 16-row tile (2 loads of 8 rows x 128 B: whole lines), three tiles ahead of the tile being finished, and goes through the upper half of the
// staging window into the accumulator layout.  Why: loads that seed the accumulators must be complete before a
// tile's first MFMA, and memory operations retire in order, so the tile start waited for the previous tile's store acknowledgements and
// for its own 128 KB of rows -- 13-17 k of a 55 k-cycle tile period at N = K = 512 (profiles/r04_clock_probe.txt; a residual-free GEMM of
// that size runs 121 us against 188, profiles/r05_residual_form_probe.txt).  Here nothing waits at the tile start, and the rows' latency
// hides behind the epilogue's own arithmetic.  The sum is (acc + bias) + x, the reference's own order (x + Linear(.), mage_model.py:48,52).
// With AFFINE (the padded-taps convolutions) the residual rows follow the convolution's own row map (optionally at half resolution, res_half).
template <int ACT, typename OT, int MT, bool AFFINE = false, int LN = LN_NONE, int OSPL = 0,
          typename HT = std::conditional_t<sizeof(OT) == 2, OT, unsigned short>, bool RESE = false>
__device__ __forceinline__ void epilogue_lean(const mage_gemm_desc& d, const f32x4 (&bias)[4], f32x4 (&acc)[MT][4], int m0, int n0,
                                              int lane, int plane, char* stg, long ysplit, const LnConsume* lnc = nullptr) {
    constexpr bool F32 = sizeof(OT) == 4;
    static_assert((LN != LN_DUAL && LN != LN_GELUBWD) || std::is_same<HT, unsigned short>::value, "training forms: bf16 rows");
    static_assert(!RESE || (sizeof(OT) == 2 && OSPL == 0 && (LN == LN_NONE || LN == LN_PRODUCE)), "residual in the epilogue: 16-bit rows out");
    constexpr int RB = F32 ? 256 : 128;            // bytes of one staged row (64 columns)
    constexpr int NCH = RB / 16;                   // 16-byte chunks per row: 16 | 8
    constexpr int RPI = 64 / NCH;                  // rows per store instruction: 4 | 8
    constexpr int NST = 16 / RPI;                  // store instructions per 16-row tile: 4 | 2
    constexpr int CPC = 16 / (int)sizeof(OT);      // columns per chunk: 4 | 8
    const int l15 = lane & 15, grp = lane >> 4;
    // write side: lane (l15, grp) owns row l15, columns nt*16 + grp*4 + {0..3} of each 16-column block nt
    int woff[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
        woff[nt] = OSPL ? l15 * RB + (((nt * 2 + (grp >> 1)) ^ l15) << 4) + (grp & 1) * 8     // hi chunk; the lo chunk is this ^ 128
                 : F32  ? l15 * RB + (((nt * 4 + grp) ^ l15) << 4)
                        : l15 * RB + (((nt * 2 + (grp >> 1)) ^ ((l15 >> 1) & 7)) << 4) + (grp & 1) * 8;
    // read side: lane -> (row rr + RPI*i, chunk cc)
    const int rr = lane / NCH, cc = lane % NCH;
    int roff[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int r = rr + RPI * i;
        roff[i] = F32 ? r * RB + ((cc ^ r) << 4) : r * RB + ((cc ^ ((r >> 1) & 7)) << 4);
    }
    const int col = n0 + cc * CPC;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)                  // the wave's private 4 KiB staging window (split rows: two 8-byte pieces 128 bytes apart)
        MAGE_DASSERT(woff[nt] >= 0 && (OSPL ? (woff[nt] | 128) + 8 : woff[nt] + (F32 ? 16 : 8)) <= 4096);
#pragma unroll
    for (int i = 0; i < NST; ++i) MAGE_DASSERT(roff[i] >= 0 && roff[i] + 16 <= 4096);
    const bool simple_rows = d.out_h == 1 && d.out_w >= d.M;
    const bool affine_rows = AFFINE && d.out_h == 1 && d.y_mul_x == 1 && d.out_w % 256 == 0;
    const bool interior = (simple_rows || affine_rows) && m0 + MT * 16 <= d.M && n0 + 64 <= d.N;     // wave-uniform
    const bool cv_ok = col < d.N;                                                    // N % 8 == 0: a chunk is all in or all out
    const long row_shift = (AFFINE && !simple_rows && affine_rows) ? (long)(m0 / d.out_w) * (d.y_img_stride - d.out_w) : 0;
    const int ldy = OSPL ? d.ldy >> 1 : d.ldy;          // split rows: ldy counts 16-bit elements, a logical element is 4 bytes
    OT* yp = (OT*)d.Y + ysplit + ((long)((m0 + rr) * d.y_mul_x + d.y_off) + row_shift) * ldy + col;      // row m0 + rr, then steps
    const long step = (long)RPI * d.y_mul_x * ldy;
    // LN_PRODUCE (fp32 out): the bf16 copy of the same rows (8 bytes per lane: 4 rows x 128 B per instruction)
    [[maybe_unused]] HT* y2p = nullptr;
    [[maybe_unused]] long step2 = 0;
    if constexpr (LN == LN_PRODUCE) {
        y2p = (HT*)d.y2 + (long)((m0 + rr) * d.y_mul_x + d.y_off) * d.ldy2 + col;
        step2 = (long)RPI * d.y_mul_x * d.ldy2;
    }

    // LN_GELUBWD: the pre-activation rows in the MFMA layout (8 bytes per lane per 16-column block), requested one 16-row tile ahead
    // requested as ROWS (2 loads of 8 rows x 128 B per 16-row tile: whole lines, finding 47) and brought into the accumulator layout
    // through the upper half of the staging window (bf16 output rows use the lower 2 KiB); host: plain rows, M and N multiples of 256
    [[maybe_unused]] u32x4 auxr[2][2];
    [[maybe_unused]] auto aux_request = [&](int mt) {
        const unsigned short* ap = (const unsigned short*)d.y2 + (long)((m0 + mt * 16 + (lane >> 3)) * d.y_mul_x + d.y_off) * d.ldy2 + n0 + (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) auxr[mt & 1][i] = *(const u32x4*)(ap + (long)(8 * i) * d.y_mul_x * d.ldy2);
    };
    if constexpr (LN == LN_GELUBWD) aux_request(0);
    // RESE: the residual rows of 16-row tile t (2 loads of 8 rows x 128 B: whole lines), kept RD tiles ahead of the tile being finished in a
    // ring of RD register pairs (the whole block at once -- 64 registers at MT = 8 -- spilled the 8-phase kernel: 145 VGPRs to scratch);
    // rows clamped to the problem (edge tiles store predicated)
    constexpr int RD = 3;
    [[maybe_unused]] u32x4 rland[RESE ? RD : 1][2];
    [[maybe_unused]] const HT* rbase = nullptr;
    [[maybe_unused]] auto res_request = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = min(m0 + t * 16 + (lane >> 3) + 8 * i, d.M - 1);
            if constexpr (AFFINE) {
                // convolution rows (padded-taps form): the residual is a [img, out_h, out_w, N] tensor of its own -- or, res_half, that tensor at
                // half resolution (nn.Upsample of a bottleneck block's identity path folded into the convolution that adds it)
                const int img = m / plane, rem = m - img * plane;
                const int oy = rem / d.out_w, ox = rem - oy * d.out_w;
                const long rrow = d.res_half ? (long)img * (plane >> 2) + (long)(oy >> 1) * (d.out_w >> 1) + (ox >> 1) : (long)m;
                rland[t % RD][i] = *(const u32x4*)(rbase + rrow * d.ldr);
            } else {
                rland[t % RD][i] = *(const u32x4*)(rbase + (long)(m * d.y_mul_x + d.y_off) * d.ldr);
            }
        }
    };
    if constexpr (RESE) {
        const int rcol = n0 + (lane & 7) * 8;
        rbase = (const HT*)d.residual + (rcol < d.N ? rcol : 0);
#pragma unroll
        for (int t = 0; t < RD && t < MT; ++t) res_request(t);
    }
    auto stage = [&](int mt, u32x4 (&o)[NST], auto apply_act) {     // math + transpose of 16-row tile mt: results land in o[] (row-contiguous)
        [[maybe_unused]] float s1 = 0.f, s2 = 0.f;
        [[maybe_unused]] uint2 auxb[4];
        if constexpr (RESE) {                          // this tile's 16 residual rows into the accumulator layout (upper half of the window)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = (lane >> 3) + 8 * i;
                *(u32x4*)(stg + 2048 + r * 128 + (((lane & 7) ^ ((r >> 1) & 7)) << 4)) = rland[mt % RD][i];
            }
            if (mt + RD < MT) res_request(mt + RD);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                auxb[nt] = *(const uint2*)(stg + 2048 + l15 * 128 + (((nt * 2 + (grp >> 1)) ^ ((l15 >> 1) & 7)) << 4) + (grp & 1) * 8);
        }
        if constexpr (LN == LN_GELUBWD) {
            if (mt + 1 < MT) aux_request(mt + 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = (lane >> 3) + 8 * i;
                *(u32x4*)(stg + 2048 + r * 128 + (((lane & 7) ^ ((r >> 1) & 7)) << 4)) = auxr[mt & 1][i];
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                auxb[nt] = *(const uint2*)(stg + 2048 + l15 * 128 + (((nt * 2 + (grp >> 1)) ^ ((l15 >> 1) & 7)) << 4) + (grp & 1) * 8);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 v;
            if constexpr (LN == LN_CONSUME) v = (acc[mt][nt] - lnc->s[nt] * lnc->mean[mt]) * lnc->rstd[mt] + bias[nt];
            else v = acc[mt][nt] + bias[nt];
            if constexpr (RESE) v = v + widen4<HT>(auxb[nt]);
            if constexpr (LN == LN_PRODUCE) {
                s1 += (v[0] + v[1]) + (v[2] + v[3]);
                s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
            if constexpr (LN == LN_GELUBWD) {              // v *= s (1 + 1.702 x (1 - s)), s = sigmoid(1.702 x)
                const uint2 ax = auxb[nt];
                const f32x4 x = f32x4{__uint_as_float(ax.x << 16), __uint_as_float(ax.x & 0xffff0000u), __uint_as_float(ax.y << 16),
                                      __uint_as_float(ax.y & 0xffff0000u)};
                const f32x4 t = x * -2.4554669596f;
                f32x4 sg;
#pragma unroll
                for (int e = 0; e < 4; ++e) sg[e] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t[e]));
                v = v * (sg * (1.0f + (x * 1.702f) * (1.0f - sg)));
            }
            if constexpr (!decltype(apply_act)::value) {
            } else if constexpr (ACT == MAGE_ACT_QUICKGELU) {
                // x * sigmoid(1.702 x) = x / (1 + 2^(-1.702 log2(e) x)): the two transcendentals (quarter rate) are the cost;
                // everything around them as 4-wide vector arithmetic, which hipcc packs into v_pk_* (one multiply fewer per
                // element than the scalar form, which scales by -1.702 and by log2(e) separately)
                const f32x4 t = v * -2.4554669596f;
                f32x4 e4;
#pragma unroll
                for (int e = 0; e < 4; ++e) e4[e] = __builtin_amdgcn_exp2f(t[e]);
                e4 = e4 + 1.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) e4[e] = __builtin_amdgcn_rcpf(e4[e]);
                v = v * e4;
            } else if (ACT != MAGE_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_apply<ACT>(v[e]);
            }
            if constexpr (OSPL != 0) {
                uint2 hi, lo;
                split_pack2<OSPL>(v[0], v[1], hi.x, lo.x);
                split_pack2<OSPL>(v[2], v[3], hi.y, lo.y);
                *(uint2*)(stg + woff[nt]) = hi;
                *(uint2*)(stg + (woff[nt] ^ 128)) = lo;
            } else if constexpr (F32) {
                *(f32x4*)(stg + woff[nt]) = v;
            } else {
                *(uint2*)(stg + woff[nt]) = uint2{pack16x2<OT>(v[0], v[1]), pack16x2<OT>(v[2], v[3])};
            }
        }
        if constexpr (LN == LN_PRODUCE) {
            // the four lane groups hold four 16-column pieces of row l15: fixed-order sum, written by group 0
            s1 += __shfl_xor(s1, 16);
            s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (grp == 0) {
                const long row = (long)(m0 + mt * 16 + l15) * d.y_mul_x + d.y_off;
                float* pp = d.ln_part + ((long)(n0 >> 6) * d.ln_part_rows + row) * 2;       // slice-major: the tile's 16 rows = 128 contiguous bytes
                *(float2*)pp = float2{s1, s2};
            }
        }
#pragma unroll
        for (int i = 0; i < NST; ++i) o[i] = *(const u32x4*)(stg + roff[i]);
    };
    auto store = [&](int mt, const u32x4 (&o)[NST]) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            // streaming (non-temporal) stores: the output is not re-read by this kernel, keep the XCD's L2 for the
            // activation panels and W that the neighbouring workgroups re-read
            if (interior) {
#ifdef MAGE_EPI_NO_STORE                               // tuning build: the epilogue's arithmetic and LDS transposition without its global stores
                asm volatile("" ::"v"(o[i]), "v"(yp));
#else
#ifdef MAGE_EPI_PLAIN_STORE                            // tuning build: cached instead of streaming stores (does the Infinity Cache keep the rows?)
                *(u32x4*)yp = o[i];
#else
                __builtin_nontemporal_store(o[i], (u32x4*)yp);
#endif
#endif
                yp += step;
                if constexpr (LN == LN_PRODUCE && F32) {
                    const uint2 pk = uint2{pack16x2<HT>(__uint_as_float(o[i][0]), __uint_as_float(o[i][1])),
                                           pack16x2<HT>(__uint_as_float(o[i][2]), __uint_as_float(o[i][3]))};
                    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                    __builtin_nontemporal_store(u32x2_t{pk.x, pk.y}, (u32x2_t*)y2p);
                    y2p += step2;
                }
            } else {
                const int m = m0 + mt * 16 + rr + RPI * i;
                if (m < d.M && cv_ok) {
                    int yrow;
                    if (simple_rows) {
                        yrow = m * d.y_mul_x + d.y_off;
                    } else {
                        const int img = m / plane;
                        const int rem = m - img * plane;
                        const int oy = rem / d.out_w;
                        const int ox = rem - oy * d.out_w;
                        yrow = img * d.y_img_stride + oy * d.y_mul_y + ox * d.y_mul_x + d.y_off;
                    }
                    __builtin_nontemporal_store(o[i], (u32x4*)((OT*)d.Y + ysplit + (long)yrow * ldy + col));
                }
            }
        }
    };
    if constexpr (LN == LN_DUAL) {                 // interior tiles, plain rows (host check): rows of y, then rows of y2 = act(y)
        OT* yp2 = (OT*)d.y2 + (long)((m0 + rr) * d.y_mul_x + d.y_off) * d.ldy2 + col;
        const long step2d = (long)RPI * d.y_mul_x * d.ldy2;
        u32x4 oa[NST], ob[NST];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            stage(mt, oa, std::false_type{});
            stage(mt, ob, std::true_type{});
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                __builtin_nontemporal_store(oa[i], (u32x4*)yp);
                yp += step;
                __builtin_nontemporal_store(ob[i], (u32x4*)yp2);
                yp2 += step2d;
            }
        }
        return;
    }
    // skewed by one tile: the LDS round trip of tile mt+1 is in flight while tile mt's rows are stored
    u32x4 o[2][NST];
    stage(0, o[0], std::true_type{});
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (mt + 1 < MT) stage(mt + 1, o[(mt + 1) & 1], std::true_type{});
        store(mt, o[mt & 1]);
    }
}

// RB (template parameter of the kernels): the residual of x + Linear(.) is the 16-bit stream (the decoder's bf16 / f16 modes keep x in that type
// between its blocks) and the rows leave in that type too: the residual is added in the epilogue (RESE above), LN_PRODUCE writes the 16-bit rows
// as its ONLY output.

// raw barrier that LDS-DMA may stay in flight across (a __syncthreads() would drain vmcnt to 0); the empty asm
// statements keep the compiler from moving LDS accesses over it
__device__ __forceinline__ void ring_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

}  // namespace
