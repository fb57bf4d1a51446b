// Fused GEMM / implicit-GEMM convolution on CDNA4 matrix cores (see include/mage_hip.h, mage_gemm).
//
// Tile: 128 (rows of A, "m") x 128 (rows of W, "n") per 256-thread workgroup, K consumed in slabs of
// 128 bytes per row (64 bf16 / 32 fp32).  4 waves as 2(m) x 2(n); each wave owns a 64x64 sub-tile as 4x4
// MFMA 16x16 accumulators (64 fp32 accumulator registers per lane).
//
// HBM -> LDS: global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip), double buffered, one barrier per
// K slab.  An LDS-DMA writes wave-base + lane*16, so the LDS image is lane-linear: [row][8 chunks of 16 B].
// Bank conflicts on the fragment reads are removed by an XOR swizzle applied on the *source* address of
// the DMA (physical chunk p of row r holds logical chunk p ^ ((r>>1)&7)) and again on the ds_read_b128.
//
// MFMA operand roles are swapped (A-operand = W rows, B-operand = activation rows) so that each lane ends
// up with 4 consecutive output columns n of ONE output row m: the epilogue then reads bias / BN / residual
// and writes Y with 16-byte (fp32) or 8-byte (bf16) vectors, no LDS transpose.
//
// fp32 mode uses v_mfma_f32_16x16x4_f32 (exact fp32 fma chain, 1/16 of the bf16 rate); it shares the
// byte-identical LDS image, loader and epilogue with the bf16 path: only the inner MFMA differs.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int TILE_BYTES = 128 * 128;          // one operand tile: 128 rows x 128 bytes
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A tile + W tile
constexpr int LDS_BYTES = 2 * STAGE_BYTES;     // double buffered: 64 KiB -> 2 workgroups per CU

struct GemmArgs {
    mage_gemm_desc d;
    const char* zero;
    int ntiles_n;
};

template <int DT> struct TT;
template <> struct TT<MAGE_F32> { typedef float elem; static constexpr int CH = 4; };
template <> struct TT<MAGE_BF16> { typedef unsigned short elem; static constexpr int CH = 8; };

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case MAGE_ACT_RELU: return fmaxf(v, 0.f);
        case MAGE_ACT_QUICKGELU: return v / (1.f + __expf(-1.702f * v));
        case MAGE_ACT_GELU_ERF: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        case MAGE_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

template <int DT, bool GATHER>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
    typedef typename TT<DT>::elem E;
    constexpr int CH = TT<DT>::CH;
    constexpr int BK = 8 * CH;
    constexpr int ES = sizeof(E);
    const mage_gemm_desc& d = g.d;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware tile order: workgroup b runs on XCD b%8; give each XCD a contiguous run of tiles so that the
    // n-tiles sharing one activation panel hit the same L2 (bijective for any grid size).
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int tm = lid / g.ntiles_n, tn = lid - tm * g.ntiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- loader state: each wave moves 4 x (8 rows x 128 B) units of each operand tile per K slab
    const int lr = lane >> 3;            // row inside a unit
    const int lp = lane & 7;             // physical 16-byte chunk
    const char* a_row[4];                // plain mode: row base pointer (or null)
    int a_img[4], a_iy[4], a_ix[4];      // gather mode
    const char* w_row[4];
    int csw[4];                          // logical chunk this lane fetches for unit i
    const int plane = d.out_h * d.out_w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = wave * 4 + i;
        const int r = u * 8 + lr;
        csw[i] = lp ^ ((r >> 1) & 7);
        const int m = m0 + r;
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
            a_row[i] = mv ? (const char*)d.A + arow * d.lda * ES : nullptr;
        }
        const int n = n0 + r;
        w_row[i] = (n < d.N) ? (const char*)d.W + (long)n * d.K * ES : nullptr;
    }

    auto issue = [&](int kt, int stage) {
        char* sa = smem + stage * STAGE_BYTES;
        char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = wave * 4 + i;
            const int kc = kt * BK + csw[i] * CH;
            const bool kv = kc < d.K;
            const char* src = g.zero;
            if (GATHER) {
                if (kv && a_img[i] >= 0) {
                    const int tap = kc / d.cin;
                    const int ci = kc - tap * d.cin;
                    const int ky = tap / d.taps_w;
                    const int kx = tap - ky * d.taps_w;
                    const int iy = a_iy[i] + ky * d.dys;
                    const int ix = a_ix[i] + kx * d.dxs;
                    if ((unsigned)iy < (unsigned)d.in_h && (unsigned)ix < (unsigned)d.in_w)
                        src = (const char*)d.A + ((long)(a_img[i] + iy * d.in_w + ix) * d.lda + ci) * ES;
                }
            } else {
                if (kv && a_row[i]) src = a_row[i] + (long)kc * ES;
            }
            glds16(src, sa + u * 1024);
            const char* wsrc = (kv && w_row[i]) ? w_row[i] + (long)kc * ES : g.zero;
            glds16(wsrc, sw + u * 1024);
        }
    };

    // ---- compute state
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, grp = lane >> 4;
    const int rsw = (l15 >> 1) & 7;                    // ((row>>1)&7) for every fragment row of this lane
    const int xoff = (wm * 64 + l15) * 128;            // + mt*16*128
    const int woff = TILE_BYTES + (wn * 64 + l15) * 128;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (d.K + BK - 1) / BK;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                               // slab kt has landed; slab kt-1's readers are done
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int pc = ((grp + 4 * t) ^ rsw) * 16;
            u32x4 xf[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xf[i] = *(const u32x4*)(st + xoff + i * 2048 + pc);
                wf[i] = *(const u32x4*)(st + woff + i * 2048 + pc);
            }
            if (DT == MAGE_BF16) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, wf[nt]), __builtin_bit_cast(bf16x8, xf[mt]), acc[mt][nt], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                __uint_as_float(wf[nt][j]), __uint_as_float(xf[mt][j]), acc[mt][nt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane holds, per (mt, nt), output row m = ..+l15 and 4 consecutive columns n = ..+grp*4
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + l15;
        if (m >= d.M) continue;
        const int img = m / plane;
        const int rem = m - img * plane;
        const int oy = rem / d.out_w;
        const int ox = rem - oy * d.out_w;
        const long yrow = (long)img * d.y_img_stride + (long)oy * d.y_mul_y + (long)ox * d.y_mul_x + d.y_off;
        const float* radd = d.rowadd ? d.rowadd + ((yrow / d.rowadd_div) % d.rowadd_mod) * (long)d.N : nullptr;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + nt * 16 + grp * 4;
            if (n >= d.N) continue;
            f32x4 v = acc[mt][nt];
            if (d.bias) v += *(const f32x4*)(d.bias + n);
            if (d.scale) v = v * *(const f32x4*)(d.scale + n) + *(const f32x4*)(d.shift + n);
            if (d.act != MAGE_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], d.act);
            }
            if (radd) v += *(const f32x4*)(radd + n);
            if (d.residual) {
                if (d.res_dtype == MAGE_F32) v += load4((const float*)d.residual + yrow * d.ldr + n);
                else v += load4((const unsigned short*)d.residual + yrow * d.ldr + n);
            }
            if (d.post_relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (d.y_dtype == MAGE_F32) store4((float*)d.Y + yrow * d.ldy + n, v);
            else store4((unsigned short*)d.Y + yrow * d.ldy + n, v);
        }
    }
}

template <int DT, bool GATHER>
int launch(const mage_gemm_desc* d, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<DT, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    GemmArgs a;
    a.d = *d;
    a.zero = (const char*)mage_zero_page();
    const int tiles_m = (d->M + BM - 1) / BM;
    a.ntiles_n = (d->N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_kernel<DT, GATHER>), dim3(tiles_m * a.ntiles_n), dim3(256), LDS_BYTES, s, a);
    MAGE_CHECK_LAUNCH("mage_gemm");
    return MAGE_OK;
}

}  // namespace

extern "C" int mage_gemm(const mage_gemm_desc* d, void* stream) {
    MAGE_CHECK_ARG(d != nullptr, "mage_gemm: null descriptor");
    MAGE_CHECK_ARG(mage_zero_page() != nullptr, "mage_gemm: mage_init() has not been called");
    MAGE_CHECK_ARG(d->dtype == MAGE_F32 || d->dtype == MAGE_BF16, "mage_gemm: bad dtype %d", d->dtype);
    MAGE_CHECK_ARG(d->y_dtype == MAGE_F32 || d->y_dtype == MAGE_BF16, "mage_gemm: bad y_dtype %d", d->y_dtype);
    MAGE_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "mage_gemm: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
    MAGE_CHECK_ARG(d->A && d->W && d->Y, "mage_gemm: null operand");
    const int ch = d->dtype == MAGE_BF16 ? 8 : 4;
    MAGE_CHECK_ARG(d->N % 4 == 0, "mage_gemm: N=%d must be a multiple of 4", d->N);
    MAGE_CHECK_ARG(d->K % ch == 0 && d->lda % ch == 0 && d->cin % ch == 0,
                   "mage_gemm: K=%d, lda=%d, cin=%d must be multiples of %d", d->K, d->lda, d->cin, ch);
    MAGE_CHECK_ARG(d->ldy % 4 == 0 && (!d->residual || d->ldr % 4 == 0), "mage_gemm: ldy/ldr must be multiples of 4");
    MAGE_CHECK_ARG(d->taps_h >= 1 && d->taps_w >= 1 && d->K == d->taps_h * d->taps_w * d->cin,
                   "mage_gemm: K=%d != taps_h*taps_w*cin = %d*%d*%d", d->K, d->taps_h, d->taps_w, d->cin);
    MAGE_CHECK_ARG(d->out_h >= 1 && d->out_w >= 1 && d->in_h >= 1 && d->in_w >= 1, "mage_gemm: bad geometry");
    MAGE_CHECK_ARG(!d->scale == !d->shift, "mage_gemm: scale and shift must be given together");
    MAGE_CHECK_ARG(!d->rowadd || (d->rowadd_div >= 1 && d->rowadd_mod >= 1), "mage_gemm: bad rowadd div/mod");
    MAGE_CHECK_ARG((((uintptr_t)d->A | (uintptr_t)d->W | (uintptr_t)d->Y) & 15) == 0, "mage_gemm: operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const bool gather = d->taps_h * d->taps_w > 1 || d->stride != 1 || d->dy0 != 0 || d->dx0 != 0 || d->in_h != d->out_h ||
                        d->in_w != d->out_w;
    if (d->dtype == MAGE_BF16) return gather ? launch<MAGE_BF16, true>(d, s) : launch<MAGE_BF16, false>(d, s);
    return gather ? launch<MAGE_F32, true>(d, s) : launch<MAGE_F32, false>(d, s);
}
