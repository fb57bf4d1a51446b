// gemm2_kernel: an EXPERIMENT on the dominant kernels' exposed epilogue ("what bounds the headline": c_fc's LayerNorm-consuming +
// QuickGELU epilogue is 36 % of its tile on the one-wave-per-SIMD kernel, whose wave owns the whole accumulator file, so nothing runs under
// it).  Here the opposite trade: TWO independent workgroups per CU (4 waves each, one per SIMD, 128 x 64 outputs per wave = 128 accumulator
// registers), one 256 x 128 tile per workgroup, no persistence -- the hardware de-phases the two residents, so one workgroup's epilogue
// (VALU / transcendental) runs under the other's K loop (MFMA) on every SIMD.  The price: 1.5x the L2 -> LDS bytes per FLOP (the tile is
// half as wide) and 1.5x the fragment reads per MFMA.
// Same bits as the other GEMM kernels: per output element the K chain is v_mfma_f32_16x16x32 over k ascending in steps of 32 from 0, and
// the epilogue is epilogue_lean (gemm_shared.h).
// NOT part of libmage_hip.so: it was built into the library for one commit ("gemm2 experiment ...") to be measured through the C ABI
// (tools/gemm2_probe.py at that commit; results: profiles/r05_gemm2_experiment.txt: bit-identical, 17-28 % slower) and then moved here.
// To re-run: copy to mage_amd/csrc/gemm2.hip, add it to SRCS in the Makefile, an `int gemm_2wg` option (common.h / runtime.hip / config.py)
// and `if (const int r = mage_gemm2_try(d, s)) return r < 0 ? r : MAGE_OK;` in front of mage_gemm4_try in gemm.hip.
//
// LDS: 3-stage ring of K slabs of 32 (A 256 rows x 64 B + W 128 rows x 64 B = 24 KB per stage, 72 KB per workgroup); 16-byte chunk c of
// row r sits at chunk c ^ h[(r >> 2) & 3], h = {0, 3, 2, 1}: conflict-free for the ds_read_b128 lane groups.  The ring's first 16 KB are
// the epilogue's four staging windows once the K loop is done.
#include "gemm_shared.h"

namespace {

struct Gemm2Args {
    mage_gemm_desc d;
    int ntiles_n, ntiles;
};

constexpr int G2_BM = 256, G2_BN = 128, G2_A = G2_BM * 64, G2_STAGE = G2_A + G2_BN * 64, G2_NST = 3, G2_LDS = G2_NST * G2_STAGE;

template <int ACT, int LN, bool HF>
__global__ __launch_bounds__(256, 2) void gemm2_kernel(const Gemm2Args g) {
    typedef std::conditional_t<HF, f16_t, unsigned short> H16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const mage_gemm_desc& d = g.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroup b runs on XCD b % 8: each XCD takes a contiguous chunk of the tile list (tn fastest), so the column tiles that share an A panel
    // follow each other in one XCD's L2
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int q8 = g.ntiles >> 3, r8 = g.ntiles & 7;
    const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    if (li >= cnt) return;
    const int tile = chunk0 + li;
    const int tm = tile / g.ntiles_n, tn = tile - tm * g.ntiles_n;
    const int nk = d.K >> 5;

    // ---- loader: a DMA = 16 rows x 64 B; A has 16 per stage (4 per wave), W 8 (2 per wave).  lane -> row lane >> 2, physical chunk lane & 3
    const int lrow = lane >> 2;
    const int hperm = (0x1230 >> (4 * ((lrow >> 2) & 3))) & 3;     // h = {0, 3, 2, 1}
    const int lch = ((lane & 3) ^ hperm) * 8;                        // logical chunk -> element offset inside the 32-wide slab
    const unsigned short* asrc[4];
    const unsigned short* wsrc[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) asrc[i] = (const unsigned short*)d.A + (long)(tm * G2_BM + (wave * 4 + i) * 16 + lrow) * d.lda + lch;
#pragma unroll
    for (int i = 0; i < 2; ++i) wsrc[i] = (const unsigned short*)d.W + (long)(tn * G2_BN + (wave * 2 + i) * 16 + lrow) * d.ldw + lch;
    auto issue = [&](int kt, int stage) {
        char* sa = smem + stage * G2_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(asrc[i] + kt * 32, sa + (wave * 4 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(wsrc[i] + kt * 32, sa + G2_A + (wave * 2 + i) * 1024);
    };

    // ---- compute: wave (wr, wc) owns rows wr*128 + [0, 128) x columns wc*64 + [0, 64)
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, grp = lane >> 4;
    const int fch = (grp ^ ((0x1230 >> (4 * ((l15 >> 2) & 3))) & 3)) << 4;       // this lane's 16-byte chunk of fragment row l15 (+ 16 mt: same (r >> 2) & 3)
    const int xoff = (wr * 128 + l15) * 64 + fch, woff = G2_A + (wc * 64 + l15) * 64 + fch;
    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int m0 = tm * G2_BM + wr * 128, n0 = tn * G2_BN + wc * 64;
    f32x4 biasm[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) biasm[b] = d.bias ? *(const f32x4*)(d.bias + n0 + b * 16 + grp * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] LnConsume lnc;
    if constexpr (LN == LN_CONSUME) {
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const float2 st = *(const float2*)(d.ln_stats + 2 * (long)(m0 + a * 16 + l15));
            lnc.mean[a] = st.x;
            lnc.rstd[a] = st.y;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) lnc.s[b] = *(const f32x4*)(d.ln_colsum + n0 + b * 16 + grp * 4);
    }
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // slab kt has landed: its 6 DMAs are older than slab kt+1's 6 (memory operations retire in order)
        if (kt + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0F76);        // vmcnt(6)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
        ring_barrier();                                             // everyone's share is in LDS; everyone has left the stage refilled below
        const char* st = smem + stage * G2_STAGE;
        u32x4 wf[4], xf[8];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const u32x4*)(st + woff + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) xf[mt] = *(const u32x4*)(st + xoff + mt * 1024);
        if (kt + 2 < nk) issue(kt + 2, stage == 0 ? 2 : stage - 1);
        __builtin_amdgcn_s_setprio(1);                              // the MFMA section ahead of the co-resident workgroup's loads / epilogue
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16x16x32<H16>(wf[nt], xf[mt], acc[mt][nt]);
        __builtin_amdgcn_s_setprio(0);
        stage = stage == 2 ? 0 : stage + 1;
    }
    ring_barrier();                                                 // the ring is free: its head becomes the four staging windows
    char* stg = smem + wave * 4096;
    const int plane = d.out_h * d.out_w;
    if constexpr (LN == LN_CONSUME) epilogue_lean<ACT, H16, 8, false, LN_CONSUME>(d, biasm, acc, m0, n0, lane, plane, stg, 0L, &lnc);
    else epilogue_lean<ACT, H16, 8, false, LN_NONE>(d, biasm, acc, m0, n0, lane, plane, stg, 0L);
}

template <int ACT, int LN, bool HF>
int launch2(const mage_gemm_desc* d, hipStream_t s) {
    const int dev = mage_device_index();
    if (dev < 0) return 0;
    static bool attr[MAGE_MAX_DEVICES] = {false};
    if (!attr[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm2_kernel<ACT, LN, HF>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
        attr[dev] = true;
    }
    Gemm2Args a;
    a.d = *d;
    a.ntiles_n = d->N / G2_BN;
    a.ntiles = (d->M / G2_BM) * a.ntiles_n;
    const int grid = (a.ntiles + 7) & ~7;
    hipLaunchKernelGGL((gemm2_kernel<ACT, LN, HF>), dim3(grid), dim3(256), G2_LDS, s, a);
    MAGE_CHECK_LAUNCH("mage_gemm");
    return 1;
}

}  // namespace

// 1 = launched, 0 = not eligible / switched off.  Eligible (option gemm_2wg set): what mage_gemm4_try takes for the generation path -- 16-bit plain
// GEMMs with the bias or LayerNorm-consuming epilogue, act none | QuickGELU, 16-bit rows out, M % 256 == 0, N % 128 == 0, K % 32 == 0.
int mage_gemm2_try(const mage_gemm_desc* d, hipStream_t s) {
    if (!mage_options().gemm_2wg) return 0;
    const bool hf = d->dtype == MAGE_F16;
    if ((d->dtype != MAGE_BF16 && !hf) || d->n_split > 1 || d->y_dtype != d->dtype) return 0;
    if (d->M % 256 || d->N % 128 || d->K % 32 || d->K < 64) return 0;
    if (d->taps_h * d->taps_w != 1 || d->stride != 1 || d->dy0 || d->dx0 || d->in_h != d->out_h || d->in_w != d->out_w || d->a_half || d->a_relu) return 0;
    if (d->out_h != 1 || d->out_w < d->M || d->y_mul_x != 1 || d->y_off != 0 || d->a_off != 0) return 0;
    if (d->scale || d->rowadd || d->post_relu || d->res_half || d->residual || d->ln_part || d->y2 || d->head_w) return 0;
    if (!d->bias || (d->ln_stats != nullptr) != (d->ln_colsum != nullptr)) return 0;
    if (d->act != MAGE_ACT_NONE && d->act != MAGE_ACT_QUICKGELU) return 0;
    if (d->ldy % 8 || d->lda % 8 || d->ldw % 8 || (((uintptr_t)d->bias | (uintptr_t)d->ln_colsum) & 15) || (((uintptr_t)d->ln_stats) & 7)) return 0;
    if (((long)d->M * d->lda + (long)d->N * d->ldw) * 2 >= (1L << 40)) return 0;
    if (hf) {
        if (d->ln_stats) return d->act == MAGE_ACT_NONE ? launch2<MAGE_ACT_NONE, LN_CONSUME, true>(d, s) : launch2<MAGE_ACT_QUICKGELU, LN_CONSUME, true>(d, s);
        return d->act == MAGE_ACT_NONE ? launch2<MAGE_ACT_NONE, LN_NONE, true>(d, s) : launch2<MAGE_ACT_QUICKGELU, LN_NONE, true>(d, s);
    }
    if (d->ln_stats) return d->act == MAGE_ACT_NONE ? launch2<MAGE_ACT_NONE, LN_CONSUME, false>(d, s) : launch2<MAGE_ACT_QUICKGELU, LN_CONSUME, false>(d, s);
    return d->act == MAGE_ACT_NONE ? launch2<MAGE_ACT_NONE, LN_NONE, false>(d, s) : launch2<MAGE_ACT_QUICKGELU, LN_NONE, false>(d, s);
}
