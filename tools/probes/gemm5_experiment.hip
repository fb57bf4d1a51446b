// gemm5_kernel: EXPERIMENT 2 on the exposed epilogue of the K = 512 GEMMs (c_fc, QKV): one wave per SIMD as in gemm4,
// but the wave's 128 x 128 block is two halves of 64 rows (2 x 128 accumulator registers) that alternate: while half h of tile t accumulates
// over its 8 K slabs, the OTHER half's finished accumulators go through the epilogue, one (16 rows x 64 columns) piece per slab -- the
// piece's VALU / transcendental work and its LDS-staged stores are in the same straight-line code as the slab's 64 MFMAs, for the compiler
// to interleave.  Every epilogue piece is epilogue_lean<.., MT = 1, ..> and every K chain is v_mfma_f32_16x16x32 over k ascending: the same
// bits as the shipped kernels.
// NOT part of libmage_hip.so (see tools/probes/gemm2_experiment.hip for how an experiment is built in; this one needs an option gemm_5 and
// `if (const int r = mage_gemm5_try(d, s)) return r < 0 ? r : MAGE_OK;` in front of mage_gemm4_try; harness: tools/gemm2_probe.py 262144 gemm_5).
// MEASURED (round 5, one MI355X, profiles/r05_gemm2_experiment.txt): bit-identical to the shipped kernels, but as compiler-scheduled code it does
// not work: 124-170 VGPRs spill at the 512-register budget (2 x 128 accumulators + fragments + the epilogue's working set), the fragment reads are
// not pipelined and the pieces are not interleaved with the MFMAs: c_fc 1037 us against 580, an epilogue-free GEMM 823 against 457.  The idea
// needs gemm4's hand-allocated registers and hand-placed instructions; this file records the structure (steps, ring, counted waits, pieces).
// LDS: 3-stage ring of 64-wide slabs, a stage = the pass's 128 A rows + the tile's 256 W rows (48 KB), + 4 x 4 KB staging = 160 KB.
#include "gemm_shared.h"

namespace {

struct Gemm5Args {
    mage_gemm_desc d;
    const char* zero;
    int ntiles_n, ntiles;
};

constexpr int G5_A = 128 * 128, G5_STAGE = G5_A + 256 * 128, G5_RING = 3 * G5_STAGE, G5_LDS = G5_RING + 4 * 4096;

template <int ACT, int LN, bool HF>
__global__ __launch_bounds__(256) void gemm5_kernel(const Gemm5Args g) {
    typedef std::conditional_t<HF, f16_t, unsigned short> H16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const mage_gemm_desc& d = g.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwg8 = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int q8 = g.ntiles >> 3, r8 = g.ntiles & 7;
    const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int chunk1 = chunk0 + q8 + (xcd < r8 ? 1 : 0);
    const int tile0 = chunk0 + li;
    if (tile0 >= chunk1) return;
    const int n_my = (chunk1 - tile0 + nwg8 - 1) / nwg8;           // tiles of this workgroup
    const int n_steps = n_my * 16;                                 // (tile, half, slab) steps

    // ---- loader: a DMA unit = 8 rows x 128 B.  A: 16 units per slab (4 per wave), W: 32 (8 per wave)
    const int lr = lane >> 3, lp = lane & 7;
    int ach[4], wch[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) ach[i] = (lp ^ ((((wave * 4 + i) * 8 + lr) >> 1) & 7)) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) wch[i] = (lp ^ ((((wave * 8 + i) * 8 + lr) >> 1) & 7)) * 8;
    auto issue = [&](int q) {                                       // step q = (tile_seq * 2 + half) * 8 + slab
        const int ts = q >> 4, h = (q >> 3) & 1, s = q & 7;
        const int tile = tile0 + ts * nwg8;
        const int tm = tile / g.ntiles_n, tn = tile - tm * g.ntiles_n;
        char* sa = smem + (q % 3) * G5_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ar = (wave * 4 + i) * 8 + lr;                 // LDS row 0..127 = (wm, row-in-half): wm = ar >> 6
            const long grow = (long)tm * 256 + (ar >> 6) * 128 + h * 64 + (ar & 63);
            glds16((const unsigned short*)d.A + grow * d.lda + s * 64 + ach[i], sa + (wave * 4 + i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long wrow = (long)tn * 256 + (wave * 8 + i) * 8 + lr;
            glds16((const unsigned short*)d.W + wrow * d.ldw + s * 64 + wch[i], sa + G5_A + (wave * 8 + i) * 1024);
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, grp = lane >> 4;
    const int rsw = (l15 >> 1) & 7;
    const int pcs[2] = {((grp + 0) ^ rsw) * 16, ((grp + 4) ^ rsw) * 16};
    const int xoff = (wm * 64 + l15) * 128, woff = G5_A + (wn * 128 + l15) * 128;
    char* stg = smem + G5_RING + wave * 4096;
    const int plane = d.out_h * d.out_w;
    f32x4 acc[2][4][2][4];                                          // [half][16-row tile][64-column piece][16-column block]

    // epilogue piece c (0..7) of half H of the tile (tm, tn): 16 rows x 64 columns through epilogue_lean<MT = 1>
    auto piece = [&](auto H_, auto C_, int tm, int tn) __attribute__((always_inline)) {
        constexpr int H = decltype(H_)::value, c = decltype(C_)::value, p = c >> 2, mt = c & 3;
        const int m0 = tm * 256 + wm * 128 + H * 64 + mt * 16, n0 = tn * 256 + wn * 128 + p * 64;
        f32x4 biasm[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) biasm[b] = *(const f32x4*)(d.bias + n0 + b * 16 + grp * 4);
        f32x4 (&a1)[1][4] = *(f32x4 (*)[1][4])&acc[H][mt][p];
        if constexpr (LN == LN_CONSUME) {
            LnConsume lnc;
            const float2 st = *(const float2*)(d.ln_stats + 2 * (long)(m0 + l15));
            lnc.mean[0] = st.x;
            lnc.rstd[0] = st.y;
#pragma unroll
            for (int b = 0; b < 4; ++b) lnc.s[b] = *(const f32x4*)(d.ln_colsum + n0 + b * 16 + grp * 4);
            epilogue_lean<ACT, H16, 1, false, LN_CONSUME>(d, biasm, a1, m0, n0, lane, plane, stg, 0L, &lnc);
        } else {
            epilogue_lean<ACT, H16, 1, false, LN_NONE>(d, biasm, a1, m0, n0, lane, plane, stg, 0L);
        }
    };

    // one K slab of half H (64 MFMAs) with, if EPI, epilogue piece S of the other half in the same straight-line code
    auto slab = [&](auto H_, auto S_, auto EPI_, int q, int etm, int etn) __attribute__((always_inline)) {
        constexpr int H = decltype(H_)::value, S = decltype(S_)::value;
        constexpr bool EPI = decltype(EPI_)::value;
        // slab q has landed: at least the 12 DMAs of slab q+1 are younger (memory operations retire in order; stores and small loads only add)
        // (with an epilogue piece in the previous slab of this pass its 11 loads / stores are younger too: vmcnt(23) keeps the prefetch two slabs deep)
        if (q + 1 >= n_steps) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        else if (EPI && S > 0) __builtin_amdgcn_s_waitcnt(0x4F77);  // vmcnt(23)
        else __builtin_amdgcn_s_waitcnt(0x0F7C);                    // vmcnt(12)
        ring_barrier();
        if (q + 2 < n_steps) issue(q + 2);
        const char* st = smem + (q % 3) * G5_STAGE;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            u32x4 wf[8], xf[4];                                     // one k-step's fragments at a time: 2 x 128 accumulators leave room for no more
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) wf[nt] = *(const u32x4*)(st + woff + nt * 2048 + pcs[t]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xf[mt] = *(const u32x4*)(st + xoff + mt * 2048 + pcs[t]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) {
                    f32x4& a = acc[H][mt][nt >> 2][nt & 3];
                    if (S == 0 && t == 0) a = mfma16x16x32<H16>(wf[nt], xf[mt], f32x4{0.f, 0.f, 0.f, 0.f});
                    else a = mfma16x16x32<H16>(wf[nt], xf[mt], a);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (EPI) piece(std::integral_constant<int, 1 - H>{}, S_, etm, etn);
    };
    auto pass = [&](auto H_, auto EPI_, int q0, int etm, int etn) __attribute__((always_inline)) {
        slab(H_, std::integral_constant<int, 0>{}, EPI_, q0 + 0, etm, etn);
        slab(H_, std::integral_constant<int, 1>{}, EPI_, q0 + 1, etm, etn);
        slab(H_, std::integral_constant<int, 2>{}, EPI_, q0 + 2, etm, etn);
        slab(H_, std::integral_constant<int, 3>{}, EPI_, q0 + 3, etm, etn);
        slab(H_, std::integral_constant<int, 4>{}, EPI_, q0 + 4, etm, etn);
        slab(H_, std::integral_constant<int, 5>{}, EPI_, q0 + 5, etm, etn);
        slab(H_, std::integral_constant<int, 6>{}, EPI_, q0 + 6, etm, etn);
        slab(H_, std::integral_constant<int, 7>{}, EPI_, q0 + 7, etm, etn);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    issue(0);
    issue(1);
    int ptm = 0, ptn = 0;
    for (int ts = 0; ts < n_my; ++ts) {
        const int tile = tile0 + ts * nwg8;
        const int tm = tile / g.ntiles_n, tn = tile - tm * g.ntiles_n;
        if (ts == 0) pass(I0{}, std::false_type{}, ts * 16, 0, 0);
        else pass(I0{}, std::true_type{}, ts * 16, ptm, ptn);       // half 1 of the previous tile leaves under half 0 of this one
        pass(I1{}, std::true_type{}, ts * 16 + 8, tm, tn);          // half 0 of this tile leaves under its half 1
        ptm = tm;
        ptn = tn;
    }
    // the last half: its eight pieces, nothing to hide them under
    piece(I1{}, std::integral_constant<int, 0>{}, ptm, ptn);
    piece(I1{}, std::integral_constant<int, 1>{}, ptm, ptn);
    piece(I1{}, std::integral_constant<int, 2>{}, ptm, ptn);
    piece(I1{}, std::integral_constant<int, 3>{}, ptm, ptn);
    piece(I1{}, std::integral_constant<int, 4>{}, ptm, ptn);
    piece(I1{}, std::integral_constant<int, 5>{}, ptm, ptn);
    piece(I1{}, std::integral_constant<int, 6>{}, ptm, ptn);
    piece(I1{}, std::integral_constant<int, 7>{}, ptm, ptn);
}

template <int ACT, int LN, bool HF>
int launch5(const mage_gemm_desc* d, hipStream_t s, int n_cu) {
    const int dev = mage_device_index();
    if (dev < 0) return 0;
    static bool attr[MAGE_MAX_DEVICES] = {false};
    if (!attr[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm5_kernel<ACT, LN, HF>, hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS);
        attr[dev] = true;
    }
    Gemm5Args a;
    a.d = *d;
    a.zero = (const char*)mage_zero_page();
    a.ntiles_n = d->N / 256;
    a.ntiles = (d->M / 256) * a.ntiles_n;
    const int grid = a.ntiles >= n_cu ? n_cu : ((a.ntiles + 7) & ~7);
    hipLaunchKernelGGL((gemm5_kernel<ACT, LN, HF>), dim3(grid), dim3(256), G5_LDS, s, a);
    MAGE_CHECK_LAUNCH("mage_gemm");
    return 1;
}

}  // namespace

// 1 = launched, 0 = not eligible / switched off (option gemm_5).  Eligible: bf16 plain GEMMs with K == 512, M and N multiples of 256, bias,
// act none | QuickGELU, optionally LayerNorm-consuming, bf16 rows out -- the decoder's QKV and c_fc.
int mage_gemm5_try(const mage_gemm_desc* d, hipStream_t s) {
    if (!mage_options().gemm_5) return 0;
    if (d->dtype != MAGE_BF16 || d->n_split > 1 || d->y_dtype != d->dtype) return 0;
    if (d->M % 256 || d->N % 256 || d->K != 512) return 0;
    if (d->taps_h * d->taps_w != 1 || d->stride != 1 || d->dy0 || d->dx0 || d->in_h != d->out_h || d->in_w != d->out_w || d->a_half || d->a_relu) return 0;
    if (d->out_h != 1 || d->out_w < d->M || d->y_mul_x != 1 || d->y_off != 0 || d->a_off != 0) return 0;
    if (d->scale || d->rowadd || d->post_relu || d->res_half || d->residual || d->ln_part || d->y2 || d->head_w) return 0;
    if (!d->bias || (d->ln_stats != nullptr) != (d->ln_colsum != nullptr)) return 0;
    if (d->act != MAGE_ACT_NONE && d->act != MAGE_ACT_QUICKGELU) return 0;
    if (d->ldy % 8 || d->lda % 8 || d->ldw % 8 || (((uintptr_t)d->bias | (uintptr_t)d->ln_colsum) & 15) || (((uintptr_t)d->ln_stats) & 7)) return 0;
    const int dev = mage_device_index();
    if (dev < 0) return 0;
    hipDeviceProp_t p;
    static int n_cu_dev[MAGE_MAX_DEVICES] = {0};
    if (!n_cu_dev[dev]) n_cu_dev[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount >= 8) ? (p.multiProcessorCount & ~7) : 256;
    if (d->ln_stats) return d->act == MAGE_ACT_NONE ? launch5<MAGE_ACT_NONE, LN_CONSUME, false>(d, s, n_cu_dev[dev]) : launch5<MAGE_ACT_QUICKGELU, LN_CONSUME, false>(d, s, n_cu_dev[dev]);
    return d->act == MAGE_ACT_NONE ? launch5<MAGE_ACT_NONE, LN_NONE, false>(d, s, n_cu_dev[dev]) : launch5<MAGE_ACT_QUICKGELU, LN_NONE, false>(d, s, n_cu_dev[dev]);
}
