// Fused transformer MLP:  x += c_proj( QuickGELU( c_fc( xn ) ) )   (mage_model.py:22-26,51: the MLP half of an
// AxialAttentionBlock, after ln_2).  bf16 MFMA, fp32 accumulation, fp32 residual stream.
//
// Why fuse: unfused, c_fc writes the hidden activation h [M, 4C] (1.07 GB at cfg2) and c_proj reads it back; both GEMMs
// have K = 512 resp. N = 512 and sit near their HBM roofline, not the MFMA one.  Here h never leaves the CU: a 64-row
// panel of xn stays in LDS, the hidden dimension is processed in chunks of 256 (c_fc chunk -> QuickGELU -> bf16 in LDS ->
// partial c_proj accumulated in registers).  HBM traffic per panel is xn in, x in/out; both weight matrices stream
// from L2 (4 MiB per 64 rows: the kernel is L2->LDS bound by construction, ~33 B/clk/CU measured on this chip).
//
// Persistent, one 512-thread workgroup per CU.  LDS (160 KiB, all of it): XN panel 64 KiB | H chunk 32 KiB | 2 W stages of
// 32 KiB (256 weight rows x 128 B).  The W stages form one continuous 2-deep ring over (panel, chunk, S1 k-slabs, S2
// pieces); the next panel's XN is fetched one 1-KiB unit per wave per iteration under the last chunk's S2 phase.
// Same lane-linear + XOR-swizzled LDS image and MFMA operand roles as gemm.hip.
#include "common.h"

namespace {

constexpr int PM = 64;                         // rows per panel
constexpr int CHN = 256;                       // hidden units per chunk
constexpr int XN_BYTES = 64 * 1024;            // panel: 64 rows x C*2 bytes (C <= 512), as C/64 slabs of [64 rows][128 B]
constexpr int H_BYTES = 32 * 1024;             // chunk: 4 slabs of [64 rows][128 B]
constexpr int W_BYTES = 32 * 1024;             // one W stage: 256 rows x 128 B
constexpr int LDS_TOTAL = XN_BYTES + H_BYTES + 2 * W_BYTES;   // 160 KiB

struct MlpArgs {
    const unsigned short* xn;                  // [M, C] bf16 (ln_2 output)
    const unsigned short* w1;                  // [4C, C] bf16 (c_fc.weight)
    const float* b1;                           // [4C]
    const unsigned short* w2;                  // [C, 4C] bf16 (c_proj.weight)
    const float* b2;                           // [C]
    float* x;                                  // [M, C] fp32 residual stream, updated in place
    const char* zero;
    int M, C;
};

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void ring_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// NT2 = 16-column MFMA tiles per wave of the c_proj output (C = NT2 * 128): 4 for C = 512, 2 for C = 256
template <int NT2>
__global__ __launch_bounds__(512, 2) void mlp_kernel(const MlpArgs g) {
    constexpr int C = NT2 * 128;
    constexpr int KS1 = C / 64;                // k-slabs of c_fc (K = C)
    constexpr int NQ = NT2 / 2;                // 256-row pieces of a c_proj k-slab (N = C)
    constexpr int S2N = 4 * NQ;                // c_proj stage loads per chunk (= KS1)
    constexpr int NCH = 4 * C / CHN;           // hidden chunks
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xn_s = smem;
    char* const h_s = smem + XN_BYTES;
    char* const w_s = smem + XN_BYTES + H_BYTES;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane >> 3, lp = lane & 7;
    const int l15 = lane & 15, grp = lane >> 4;
    const int rsw = (l15 >> 1) & 7;
    const int npanel = (g.M + PM - 1) / PM;

    // ---- loader cursor over the continuous stream of W stage loads
    int ld_panel = blockIdx.x, ld_chunk = 0, ld_idx = 0, ld_stage = 0;    // ld_idx: 0..KS1-1 = S1 slabs, KS1..KS1+S2N-1 = S2 pieces
    auto issue_stage = [&]() {
        char* dst = w_s + ld_stage * W_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = wave * 4 + i;                    // unit: 8 rows x 128 B
            const int r = u * 8 + lr;                      // row inside the 256-row stage
            const int c = lp ^ ((r >> 1) & 7);             // logical 16-byte chunk this lane fetches
            const unsigned short* src;
            if (ld_idx < KS1) {                            // c_fc: rows = hidden units of the chunk, k-slab ld_idx
                src = g.w1 + (long)(ld_chunk * CHN + r) * C + ld_idx * 64 + c * 8;
            } else {                                       // c_proj piece (s2, q): wave w's columns w*NT2*16 + q*32 + i
                const int t = ld_idx - KS1, s2 = t / NQ, q = t - s2 * NQ;
                const int n = (r >> 5) * (NT2 * 16) + q * 32 + (r & 31);
                src = g.w2 + (long)n * (4 * C) + ld_chunk * CHN + s2 * 64 + c * 8;
            }
            glds16(src, dst + u * 1024);
        }
        ld_stage ^= 1;
        if (++ld_idx == KS1 + S2N) {
            ld_idx = 0;
            if (++ld_chunk == NCH) { ld_chunk = 0; ld_panel += gridDim.x; }
        }
    };
    // one 1-KiB unit (8 rows x 128 B of one k-slab) of panel p's XN; unit ids 0 .. 8*KS1-1, wave-major
    auto issue_xn_unit = [&](int p, int j) {
        const int u = wave * KS1 + j;                      // 8*KS1 units in all
        const int slab = u >> 3, r = (u & 7) * 8 + lr;
        const int c = lp ^ ((r >> 1) & 7);
        const long m = (long)p * PM + r;
        const void* src = m < g.M ? (const void*)(g.xn + m * C + slab * 64 + c * 8) : (const void*)g.zero;
        glds16(src, xn_s + slab * 8192 + (u & 7) * 1024);
    };

    if (ld_panel >= npanel) return;
#pragma unroll
    for (int j = 0; j < KS1; ++j) issue_xn_unit(ld_panel, j);
    issue_stage();

    int c_stage = 0;
    f32x4 acc1[4][2], acc2[4][NT2];
    for (int panel = blockIdx.x; panel < npanel; panel += gridDim.x) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < NT2; ++b) acc2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < NCH; ++ch) {
            // ---------------- S1: acc1[64 x 256 chunk] = XN . W1[chunk]^T, this wave's 32 hidden units
#pragma unroll
            for (int a = 0; a < 4; ++a) { acc1[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            f32x4 bv[2];                                   // c_fc bias of this wave's 2 x 4 hidden units: lands under S1
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bv[nt] = *(const f32x4*)(g.b1 + ch * CHN + wave * 32 + nt * 16 + grp * 4);
            for (int s = 0; s < KS1; ++s) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ring_barrier();
                if (ld_panel < npanel) issue_stage();
                const char* xs = xn_s + s * 8192 + l15 * 128;
                const char* ws = w_s + c_stage * W_BYTES + (wave * 32 + l15) * 128;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int pc = ((grp + 4 * t) ^ rsw) * 16;
                    u32x4 xf[4], wf[2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) xf[i] = *(const u32x4*)(xs + i * 2048 + pc);
#pragma unroll
                    for (int i = 0; i < 2; ++i) wf[i] = *(const u32x4*)(ws + i * 2048 + pc);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc1[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, wf[nt]), __builtin_bit_cast(bf16x8, xf[mt]), acc1[mt][nt], 0, 0, 0);
                }
                c_stage ^= 1;
            }
            // ---------------- bias + QuickGELU -> bf16 -> H (k-slab layout for S2: hidden unit = k)
            {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int hcol = wave * 32 + nt * 16 + grp * 4;            // hidden unit inside the chunk (0..255)
                    const int slab = hcol >> 6, lc = (hcol & 63) >> 3;         // slab of 64, logical 16-byte chunk
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const int row = mt * 16 + l15;
                        f32x4 v = acc1[mt][nt] + bv[nt];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v[e]));
                        uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                        *(uint2*)(h_s + slab * 8192 + row * 128 + ((lc ^ ((row >> 1) & 7)) << 4) + (grp & 1) * 8) = pk;
                    }
                }
            }
            // ---------------- S2: acc2[64 x C] += H[64 x 256] . W2[:, chunk]^T, this wave's NT2*16 columns
            for (int t2 = 0; t2 < S2N; ++t2) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ring_barrier();                                // also orders the H writes above before the first S2 read
                if (ld_panel < npanel) issue_stage();
                if (ch == NCH - 1 && panel + (int)gridDim.x < npanel) issue_xn_unit(panel + gridDim.x, t2);   // S2N == KS1
                const int s2 = t2 / NQ, q = t2 - s2 * NQ;
                const char* hs = h_s + s2 * 8192 + l15 * 128;
                const char* ws = w_s + c_stage * W_BYTES + (wave * 32 + l15) * 128;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int pc = ((grp + 4 * t) ^ rsw) * 16;
                    u32x4 hf[4], wf[2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) hf[i] = *(const u32x4*)(hs + i * 2048 + pc);
#pragma unroll
                    for (int i = 0; i < 2; ++i) wf[i] = *(const u32x4*)(ws + i * 2048 + pc);
#pragma unroll
                    for (int qq = 0; qq < NQ; ++qq) {          // static accumulator index: q is wave-uniform
                        if (qq == q) {
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                                for (int mt = 0; mt < 4; ++mt)
                                    acc2[mt][qq * 2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                        __builtin_bit_cast(bf16x8, wf[nt]), __builtin_bit_cast(bf16x8, hf[mt]), acc2[mt][qq * 2 + nt], 0, 0, 0);
                        }
                    }
                }
                c_stage ^= 1;
            }
        }
        // ---------------- panel epilogue: x[m, n] += acc2 + b2   (registers -> permlane16_swap -> 8 consecutive columns per lane)
        const int n0 = wave * NT2 * 16;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const long m = (long)panel * PM + mt * 16 + l15;
            const long mc = m < g.M ? m : g.M - 1;
#pragma unroll
            for (int k = 0; k < NT2 / 2; ++k) {
                const int ncol = n0 + 16 * (2 * k + (grp & 1)) + 8 * (grp >> 1);
                float* xp = g.x + mc * C + ncol;
                const f32x4 r0 = *(const f32x4*)xp, r1 = *(const f32x4*)(xp + 4);
                const f32x4 c0 = *(const f32x4*)(g.b2 + ncol), c1 = *(const f32x4*)(g.b2 + ncol + 4);
                f32x4 v0, v1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc2[mt][2 * k][e]), __float_as_uint(acc2[mt][2 * k + 1][e]),
                                                                    false, false);
                    v0[e] = __uint_as_float(r[0]);
                    v1[e] = __uint_as_float(r[1]);
                }
                if (m < g.M) {
                    __builtin_nontemporal_store(v0 + c0 + r0, (f32x4*)xp);
                    __builtin_nontemporal_store(v1 + c1 + r1, (f32x4*)xp + 1);
                }
            }
        }
    }
}

template <int NT2>
int mlp_launch(const MlpArgs& a, hipStream_t s) {
    static bool attr_set = false;
    static int n_cu = 256;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)mlp_kernel<NT2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n_cu = p.multiProcessorCount;
        attr_set = true;
    }
    const int npanel = (a.M + PM - 1) / PM;
    hipLaunchKernelGGL((mlp_kernel<NT2>), dim3(npanel < n_cu ? npanel : n_cu), dim3(512), LDS_TOTAL, s, a);
    MAGE_CHECK_LAUNCH("mage_mlp_fused");
    return MAGE_OK;
}

}  // namespace

extern "C" int mage_mlp_fused(const void* xn, const void* w_fc, const float* b_fc, const void* w_proj, const float* b_proj,
                              float* x, int64_t M, int32_t C, void* stream) {
    MAGE_CHECK_ARG(xn && w_fc && b_fc && w_proj && b_proj && x, "mage_mlp_fused: null pointer");
    MAGE_CHECK_ARG(mage_zero_page() != nullptr, "mage_mlp_fused: mage_init() has not been called");
    MAGE_CHECK_ARG(M > 0 && M < (1L << 31) && (C == 256 || C == 512), "mage_mlp_fused: M=%ld C=%d unsupported (C must be 256 or 512)",
                   (long)M, C);
    MlpArgs a{(const unsigned short*)xn, (const unsigned short*)w_fc, b_fc, (const unsigned short*)w_proj, b_proj, x,
              (const char*)mage_zero_page(), (int)M, C};
    return C == 512 ? mlp_launch<4>(a, (hipStream_t)stream) : mlp_launch<2>(a, (hipStream_t)stream);
}
