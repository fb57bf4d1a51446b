// Weight-gradient GEMM without transposed copies (training path, see include/mage_hip.h: mage_gemm_tn).
//
//     P[s][n][k] = sum over the tokens t of slice s of  dY[t][n] * X[t][k]          (bf16 operands, fp32 accumulation and output)
//
// Both operands are row-major over the CONTRACTION index (a token is a row), the opposite of what an MFMA operand fragment wants (8
// consecutive contraction elements per lane).  mage_gemm's weight-gradient route therefore wrote dY^T and X^T first (11.8 ms of a 92 ms
// training step at cfg2: 51 GB moved).  Here the tiles go to LDS as they are and the fragments are read with gfx950's transposing LDS
// load: `ds_read_b64_tr_b16` takes a per-lane address of 4 contiguous 16-bit elements; within a group of 16 lanes, lanes 4r..4r+3
// supply row r of a [4][16] block (any row pitch) and lane c receives column c -- 4 consecutive tokens of one column: half an MFMA
// operand (semantics pinned on the box with tools/probes/tr_probe.hip).
//
// Workgroup = one 256 (columns of dY) x 256 (columns of X) output tile of one token slice; 8 waves as 2 x 4, each 128 x 64 (8 x 4
// accumulators of v_mfma_f32_16x16x32_bf16), K slabs of 64 tokens in a 2-stage LDS ring filled by LDS-DMA.  LDS image of an operand's
// slab: 32 units of 1 KB = [8 tokens][64 columns] (a unit is one wave-wide DMA: 8 rows x 128 contiguous bytes from global memory), the
// four 32-byte chunks of a row XOR-swizzled with (token & 3) so that the four rows a lane group reads fall in different bank quarters.
#include <cstdlib>

#include "common.h"

int mage_gemm_tn4_try(const void* dY, int64_t lda, const void* X, int64_t ldb, int64_t T, int32_t N, int32_t K, int32_t n_split, int64_t tps,
                      float* partials, float* db_partials, hipStream_t stream);      // gemm4.hip: the one-wave-per-SIMD form of this kernel

namespace {

struct TnArgs {
    const unsigned short* A;   // dY [T, lda]
    const unsigned short* B;   // X  [T, ldb]
    float* P;                  // [n_split][N][K]
    float* DB;                 // [n_split][N] column sums of dY over the slice (bias gradient partials), or null
    const char* zero;
    long lda, ldb, T, tps;     // tps = tokens per slice (multiple of 64)
    int N, K, n_split, ntk;    // ntk = K / 256
};

typedef short tr4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16_tn(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0,
                                     0);
}
__device__ __forceinline__ tr4 lds_tr(const char* p) {
#ifdef MAGE_TN_NOTR      // tuning build: a plain 8-byte LDS read at the same address (wrong values, same traffic): is the transposing read the limiter?
    return *(const tr4*)p;
#else
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4*)p);
#endif
}

__global__ __launch_bounds__(512) void gemm_tn_kernel(const TnArgs g) {
    constexpr int PART = 32768, STAGE = 2 * PART;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x / g.n_split, s = blockIdx.x - tile * g.n_split;
    const int tm = tile / g.ntk, tk = tile - tm * g.ntk;
    const long t_begin = (long)s * g.tps;
    const long t_end = t_begin + g.tps < g.T ? t_begin + g.tps : g.T;
    const int nslab = t_end > t_begin ? (int)((t_end - t_begin + 63) >> 6) : 0;

    // ---- loader: wave w moves units 4w..4w+3 of each operand: column strip w>>1 (64 columns), token blocks (w&1)*4 + j (8 tokens each)
    const int rr = lane >> 3, pc = lane & 7;
    // logical 32-byte chunk = physical ^ (token & 3) ^ (token block & 1): the four rows a 16-lane group reads fall in four different
    // 8-bank slots, and the two groups of a 32-lane LDS pass (token blocks g, g+1: 1 KB apart, the same banks) in complementary slots
    const int strip = wave >> 1, tb0 = (wave & 1) * 4;
    const uintptr_t zero = (uintptr_t)g.zero;
    // per-lane source addresses of this wave's four units of slab 0, advanced by one slab (64 rows) per issue: no 64-bit multiplies in
    // the loop (the two waves of a SIMD share its VALU with each other's MFMA issue)
    uintptr_t pa[4], pb[4];
    long tl[4];                                        // tokens left from this lane's row to the slice's end (<= 0: past the end)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col_in_strip = (((pc >> 1) ^ (rr & 3) ^ (j & 1)) << 4) + ((pc & 1) << 3);
        const long t = t_begin + (tb0 + j) * 8 + rr;
        pa[j] = (uintptr_t)(g.A + t * g.lda + (long)tm * 256 + strip * 64 + col_in_strip);
        pb[j] = (uintptr_t)(g.B + t * g.ldb + (long)tk * 256 + strip * 64 + col_in_strip);
        tl[j] = t_end - t;
    }
    const uintptr_t da = (uintptr_t)(64 * g.lda * 2), db_ = (uintptr_t)(64 * g.ldb * 2);
    auto issue = [&](int stage) {
        char* base = smem + stage * STAGE + (wave * 4) * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // integer select: one v_cndmask pair per address (a pointer select between an SGPR-based and a VGPR-based address made hipcc
            // emit two differently addressed loads under exec-mask branches)
            const bool live = tl[j] > 0;
            glds16_tn((const void*)(live ? pa[j] : zero), base + j * 1024);
            glds16_tn((const void*)(live ? pb[j] : zero), base + PART + j * 1024);
            pa[j] += da;
            pb[j] += db_;
            tl[j] -= 64;
        }
    };

    // ---- compute state
    const int wm = wave >> 2, wn = wave & 3;
    const int i = lane & 15, grp = lane >> 4;
    // fragment byte offset inside an operand's part:  (strip*8 + t*4 + grp)*1024 + (h*4 + (i>>2))*128 + ((q ^ (i>>2))*32) + (i&3)*8
    const int lane_off = grp * 1024 + (i >> 2) * 128 + (i & 3) * 8;
    const int sw = (i >> 2) ^ (grp & 1);
    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient: the dY fragments are in registers anyway -- the waves of the first X-tile's first wave column also add them up
    // (v_dot2_f32_bf16 against (1, 1): two tokens per instruction, fp32 accumulation; 64 VALU instructions per slab beside 64 MFMAs)
    const bool do_db = g.DB != nullptr && tk == 0 && wn == 0;            // wave-uniform
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
    const bf16x2v ones2 = {(__bf16)1.0f, (__bf16)1.0f};
    float dbs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    if (nslab > 0) issue(0);
    for (int kt = 0; kt < nslab; ++kt) {
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): this wave's share of slab kt has landed
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // everyone's share is in LDS; everyone is done with the other stage
        asm volatile("" ::: "memory");
        if (kt + 1 < nslab) issue((kt + 1) & 1);
        const char* st = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x8 bf[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {           // X columns wn*64 + nt*16: strip wn, chunk nt
                const char* p = st + PART + (wn * 8 + t * 4) * 1024 + lane_off + ((nt ^ sw) << 5);
                const tr4 lo = lds_tr(p), hi = lds_tr(p + 512);
                bf[nt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {           // dY columns wm*128 + mt*16: strip wm*2 + (mt>>2), chunk mt&3
                const char* p = st + ((wm * 2 + (mt >> 2)) * 8 + t * 4) * 1024 + lane_off + (((mt & 3) ^ sw) << 5);
                const tr4 lo = lds_tr(p), hi = lds_tr(p + 512);
                const bf16x8 af = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[nt], af, acc[mt][nt], 0, 0, 0);
                if (do_db) {
                    // (the pairs are taken from the two 8-byte loads: indexing a bit-cast of the assembled operand gave hipcc's dot2 the first
                    // pair four times)
                    const uint2 w0 = __builtin_bit_cast(uint2, lo), w1 = __builtin_bit_cast(uint2, hi);
                    float d = dbs[mt];
                    d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, w0.x), ones2, d, false);
                    d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, w0.y), ones2, d, false);
                    d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, w1.x), ones2, d, false);
                    d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, w1.y), ones2, d, false);
                    dbs[mt] = d;
                }
            }
        }
    }
    // ---- epilogue: lane (i, grp) of acc[mt][nt] holds row (dY column) mt*16 + i, columns (X columns) nt*16 + grp*4 + {0..3}
    float* out = g.P + ((long)s * g.N + (long)tm * 256 + wm * 128) * g.K + (long)tk * 256 + wn * 64;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) *(f32x4*)(out + (long)(mt * 16 + i) * g.K + nt * 16 + grp * 4) = acc[mt][nt];
    if (do_db) {                                       // lane (i, grp) summed column mt*16 + i over the tokens 8*grp.. of every 32: add the four groups
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            float v = dbs[mt];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (grp == 0) g.DB[(long)s * g.N + (long)tm * 256 + wm * 128 + mt * 16 + i] = v;
        }
    }
}

// column sums of bf16 rows (bias gradients db = sum_t dY[t, :]): partial[p][c] over the p-th of n_part row chunks, fp32.  A workgroup owns
// 512 columns (64 lanes x 16 bytes) of one chunk; its four waves take the rows t = w, w+4, ... (independent 16-byte loads) and their sums are
// added in a fixed order through LDS.
__global__ __launch_bounds__(256) void colsum_kernel(const unsigned short* __restrict__ x, long ld, long T, int C, float* __restrict__ part, int n_part) {
    __shared__ f32x4 red[4][64][2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 512 + lane * 8;
    const int p = blockIdx.y;
    const long per = (T + n_part - 1) / n_part;
    const long t0 = (long)p * per, t1 = t0 + per < T ? t0 + per : T;
    f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (c < C) {
        long t = t0 + w;
        for (; t + 12 < t1; t += 16) {
            const uint4 r0 = *(const uint4*)(x + t * ld + c), r1 = *(const uint4*)(x + (t + 4) * ld + c);
            const uint4 r2 = *(const uint4*)(x + (t + 8) * ld + c), r3 = *(const uint4*)(x + (t + 12) * ld + c);
#define ACC8(r)                                                                                                          \
    s0 += f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)}; \
    s1 += f32x4{__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u), __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
            ACC8(r0) ACC8(r1) ACC8(r2) ACC8(r3)
        }
        for (; t < t1; t += 4) {
            const uint4 r0 = *(const uint4*)(x + t * ld + c);
            ACC8(r0)
        }
#undef ACC8
    }
    red[w][lane][0] = s0;
    red[w][lane][1] = s1;
    __syncthreads();
    if (w == 0 && c < C) {
        float* o = part + (long)p * C + c;
        *(f32x4*)o = (red[0][lane][0] + red[1][lane][0]) + (red[2][lane][0] + red[3][lane][0]);
        *(f32x4*)(o + 4) = (red[0][lane][1] + red[1][lane][1]) + (red[2][lane][1] + red[3][lane][1]);
    }
}

}  // namespace

extern "C" int mage_gemm_tn(const void* dY, int64_t lda, const void* X, int64_t ldb, int64_t T, int32_t N, int32_t K, int32_t n_split,
                            int64_t tokens_per_split, float* partials, float* db_partials, void* stream) {
    MAGE_CHECK_ARG(dY && X && partials && T > 0 && N > 0 && K > 0, "mage_gemm_tn: bad arguments");
    MAGE_CHECK_ARG(N % 256 == 0 && K % 256 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= N && ldb >= K,
                   "mage_gemm_tn: N=%d and K=%d must be multiples of 256, lda / ldb multiples of 8", N, K);
    MAGE_CHECK_ARG(n_split >= 1 && tokens_per_split > 0 && tokens_per_split % 64 == 0 && (int64_t)n_split * tokens_per_split >= T,
                   "mage_gemm_tn: n_split slices of tokens_per_split (a multiple of 64) tokens must cover T");
    MAGE_CHECK_ARG(((((uintptr_t)dY | (uintptr_t)X | (uintptr_t)partials)) & 15) == 0, "mage_gemm_tn: operands must be 16-byte aligned");
    MAGE_CHECK_ARG(mage_zero_page() != nullptr, "mage_gemm_tn: mage_init() has not been called");
    if (const int r = mage_gemm_tn4_try(dY, lda, X, ldb, T, N, K, n_split, tokens_per_split, partials, db_partials, (hipStream_t)stream))
        return r < 0 ? r : MAGE_OK;
    static bool attr[MAGE_MAX_DEVICES] = {false};
    const int dev = mage_device_index();
    MAGE_CHECK_ARG(dev >= 0, "mage_gemm_tn: no current device");
    if (!attr[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr[dev] = true;
    }
    TnArgs a;
    a.A = (const unsigned short*)dY;
    a.B = (const unsigned short*)X;
    a.P = partials;
    a.DB = db_partials;
    a.zero = (const char*)mage_zero_page();
    a.lda = lda;
    a.ldb = ldb;
    a.T = T;
    a.tps = tokens_per_split;
    a.N = N;
    a.K = K;
    a.n_split = n_split;
    a.ntk = K / 256;
    const long grid = (long)(N / 256) * (K / 256) * n_split;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)grid), dim3(512), 128 * 1024, (hipStream_t)stream, a);
    MAGE_CHECK_LAUNCH("mage_gemm_tn");
    return MAGE_OK;
}

extern "C" int mage_colsum(const void* x, int64_t ld, int64_t T, int32_t C, float* partials, int32_t n_part, void* stream) {
    MAGE_CHECK_ARG(x && partials && T > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0 && n_part >= 1 && (((uintptr_t)x | (uintptr_t)partials) & 15) == 0,
                   "mage_colsum: C and ld multiples of 8, 16-byte aligned operands");
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((C + 511) / 512), (unsigned)n_part), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x, (long)ld, (long)T, C, partials, n_part);
    MAGE_CHECK_LAUNCH("mage_colsum");
    return MAGE_OK;
}
