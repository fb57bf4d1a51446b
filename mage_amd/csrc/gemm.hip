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

#ifndef MAGE_ABL
#define MAGE_ABL 0
#endif
#ifndef MAGE_STAGGER
#define MAGE_STAGGER 100          // x64 clocks
#endif

#if MAGE_ABL == 4
__device__ unsigned long long mage_dbg[8 * 65536];
extern "C" int mage_debug_read(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mage_dbg), bytes);
}
#define DBG_T(slot) do { if (lane == 0 && wave == 0 && blockIdx.x < 65536) mage_dbg[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DBG_T(slot)
#endif

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

template <int ACT>
__device__ __forceinline__ float act_apply(float v) {
    if (ACT == MAGE_ACT_RELU) return fmaxf(v, 0.f);
    if (ACT == MAGE_ACT_QUICKGELU) return v * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v));
    if (ACT == MAGE_ACT_GELU_ERF) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    return v;
}

// Epilogue of one wave's 64x64 sub-tile.  The MFMA layout leaves each lane with 4 consecutive n of 16 different rows
// per accumulator; stores straight from it would touch 16 rows x 32..64 B per instruction.  Instead the wave transposes
// its sub-tile through a wave-private LDS staging tile (32 rows x 64 fp32, +4 pad: conflict-free b128 writes), in two
// halves, and reads it back row-major with VEC consecutive columns per lane: a store instruction then covers whole
// 128/256-byte lines, and the per-column vectors (bias, BN scale/shift) are loaded once per lane.
template <int ACT, int VEC, typename OT>
__device__ __forceinline__ void epilogue_rows(const mage_gemm_desc& d, f32x4 (&acc)[4][4], char* smem, int m0, int n0,
                                              int wave, int lane, int plane) {
    constexpr int LPR = 64 / VEC;          // lanes per row (16 | 8)
    constexpr int RPP = 64 / LPR;          // rows per pass (4 | 8)
    constexpr int NQ = 64 / RPP;           // row slots per lane (16 | 8)
    constexpr int NV = VEC / 4;            // f32x4 vectors per slot (1 | 2)
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, grp = lane >> 4;
    const int rsub = lane / LPR;           // row inside a pass
    const int c0 = (lane % LPR) * VEC;     // first column inside the wave's 64
    const int n = n0 + wn * 64 + c0;
    const bool nv = n < d.N;               // N % VEC == 0 is checked on the host for bf16 output
    const bool simple_rows = d.out_h == 1 && d.out_w >= d.M;       // no regrouping: yrow = m*y_mul_x + y_off
    // Rows/columns outside the problem are CLAMPED to valid ones for the loads (hipcc turns a predicated load into a
    // branch + s_waitcnt vmcnt(0) per element, serialising the round trips); only the stores are predicated.
    const int n_ld = nv ? n : 0;
    int yrow[NQ];
    unsigned valid = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int mq = m0 + wm * 64 + q * RPP + rsub;
        const int m = min(mq, d.M - 1);
        if (simple_rows) {
            yrow[q] = m * d.y_mul_x + d.y_off;
        } else {
            const int img = m / plane;
            const int rem = m - img * plane;
            const int oy = rem / d.out_w;
            const int ox = rem - oy * d.out_w;
            yrow[q] = img * d.y_img_stride + oy * d.y_mul_y + ox * d.y_mul_x + d.y_off;
        }
        valid |= (mq < d.M && nv) ? (1u << q) : 0u;
    }
    // Everything the epilogue reads from global memory is requested HERE, before the first store: vmcnt retires in
    // order, so a load issued after a store would make its wait drain that store's full round trip.
    f32x4 extra[NQ][NV];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int h = 0; h < NV; ++h) extra[q][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (d.residual) {
        if (d.res_dtype == MAGE_F32) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int h = 0; h < NV; ++h) extra[q][h] = load4((const float*)d.residual + (long)yrow[q] * d.ldr + n_ld + 4 * h);
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int h = 0; h < NV; ++h)
                    extra[q][h] = load4((const unsigned short*)d.residual + (long)yrow[q] * d.ldr + n_ld + 4 * h);
        }
    }
    if (d.rowadd) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float* rp = d.rowadd + (long)((yrow[q] / d.rowadd_div) % d.rowadd_mod) * d.N + n_ld;
#pragma unroll
            for (int h = 0; h < NV; ++h) extra[q][h] += *(const f32x4*)(rp + 4 * h);
        }
    }
    f32x4 bias4[NV], scale4[NV], shift4[NV];
#pragma unroll
    for (int h = 0; h < NV; ++h) {
        bias4[h] = d.bias ? *(const f32x4*)(d.bias + n_ld + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
        scale4[h] = d.scale ? *(const f32x4*)(d.scale + n_ld + 4 * h) : f32x4{1.f, 1.f, 1.f, 1.f};
        shift4[h] = d.scale ? *(const f32x4*)(d.shift + n_ld + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float lo = d.post_relu ? 0.f : -INFINITY;     // post-ReLU as one max (scale/shift default to 1/0: one fma)
    DBG_T(4);
    __syncthreads();                                   // all waves are done reading the last K slab
    DBG_T(5);
    float* stg = (float*)smem + wave * (32 * 68);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) *(f32x4*)(stg + (mh * 16 + l15) * 68 + nt * 16 + grp * 4) = acc[half * 2 + mh][nt];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < NQ / 2; ++p) {
            const int q = half * (NQ / 2) + p;
            f32x4 v[NV];
#pragma unroll
            for (int h = 0; h < NV; ++h) {
                v[h] = *(const f32x4*)(stg + (p * RPP + rsub) * 68 + c0 + 4 * h);
                v[h] = (v[h] + bias4[h]) * scale4[h] + shift4[h];
                if (ACT != MAGE_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[h][e] = act_apply<ACT>(v[h][e]);
                }
                v[h] += extra[q][h];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[h][e] = fmaxf(v[h][e], lo);
            }
            if (valid & (1u << q)) {
                int yr = yrow[q];
#if MAGE_ABL == 3
                yr &= 127;                             // ablation: all tiles write the same small (L2-resident) region
#endif
                OT* yp = (OT*)d.Y + (long)yr * d.ldy + n;
                if (NV == 1) store4(yp, v[0]);
                else store8(yp, v[0], v[NV - 1]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int DT, bool GATHER, int ACT>
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
#ifndef MAGE_NO_STAGGER
    // De-phase the two workgroups that share a CU.  All first-generation workgroups start together and every tile takes
    // the same time, so without this the whole chip alternates between "everyone in the MFMA loop" and "everyone
    // storing" (measured: epilogue ~= main loop).  Delaying every other first-generation workgroup by about half a main
    // loop makes one workgroup's stores overlap its neighbour's MFMAs; later generations inherit the offset.
    if (bid < 2 * 256 && ((bid >> 3) & 32)) __builtin_amdgcn_s_sleep(MAGE_STAGGER);
#endif
    DBG_T(0);
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
#if MAGE_ABL == 2
            {   // ablation: loads + LDS reads only, no MFMA
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[i][0][0] += __uint_as_float(xf[i][0] ^ wf[i][1]); }
            }
            if (false) {
#else
            if (DT == MAGE_BF16) {
#endif
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

#if MAGE_ABL == 1
    {   // ablation: main loop only (keep the accumulators alive, store nothing)
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) sacc += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
        if (sacc == 123456.789f) ((float*)d.Y)[0] = sacc;
        return;
    }
#endif
    // ---- epilogue (see epilogue_rows): fp32 output -> 4 columns per lane, bf16 output -> 8 columns per lane, so that
    // every store instruction is a 16-byte-per-lane dwordx4 (a CU retires roughly one wave-store per ~70 cycles
    // whatever its width: fewer, wider stores).
    DBG_T(1);
    if (d.y_dtype == MAGE_F32) epilogue_rows<ACT, 4, float>(d, acc, smem, m0, n0, wave, lane, plane);
    else epilogue_rows<ACT, 8, unsigned short>(d, acc, smem, m0, n0, wave, lane, plane);
#if MAGE_ABL == 4
    DBG_T(6);
    __builtin_amdgcn_s_waitcnt(0);
    DBG_T(2);
    if (lane == 0 && wave == 0 && bid < 65536) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        mage_dbg[bid * 8 + 3] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
}

template <int DT, bool GATHER, int ACT>
int launch_act(const mage_gemm_desc* d, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<DT, GATHER, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    GemmArgs a;
    a.d = *d;
    a.zero = (const char*)mage_zero_page();
    const int tiles_m = (d->M + BM - 1) / BM;
    a.ntiles_n = (d->N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_kernel<DT, GATHER, ACT>), dim3(tiles_m * a.ntiles_n), dim3(256), LDS_BYTES, s, a);
    MAGE_CHECK_LAUNCH("mage_gemm");
    return MAGE_OK;
}

template <int DT, bool GATHER>
int launch(const mage_gemm_desc* d, hipStream_t s) {
    switch (d->act) {
        case MAGE_ACT_NONE: return launch_act<DT, GATHER, MAGE_ACT_NONE>(d, s);
        case MAGE_ACT_RELU: return launch_act<DT, GATHER, MAGE_ACT_RELU>(d, s);
        case MAGE_ACT_QUICKGELU: return launch_act<DT, GATHER, MAGE_ACT_QUICKGELU>(d, s);
        case MAGE_ACT_GELU_ERF: return launch_act<DT, GATHER, MAGE_ACT_GELU_ERF>(d, s);
        default: mage_set_error("mage_gemm: activation %d is not available in the GEMM epilogue", d->act); return MAGE_EINVAL;
    }
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
    MAGE_CHECK_ARG(d->y_dtype != MAGE_BF16 || (d->N % 8 == 0 && d->ldy % 8 == 0 && (!d->residual || d->ldr % 8 == 0)),
                   "mage_gemm: bf16 output needs N, ldy (and ldr) multiples of 8 (16-byte stores)");
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
