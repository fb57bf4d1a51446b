// Codebook quantiser (nearest neighbour), embedding gathers, row argmax and cross entropy.
// Integer/index outputs here must be bit-exact against the reference, so the distance follows the
// reference FORMULA (vqvae_model.py:14-21): fl32(fl32(|c|^2 + |z|^2) - 2*<z,c>), first minimum wins.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void vq_prepare_kernel(const float* __restrict__ cb, int K, int D,
                                                         float* __restrict__ cbt, float* __restrict__ c2) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    double s = 0.0;
    for (int d = 0; d < D; ++d) {
        const float v = cb[(long)k * D + d];
        cbt[(long)d * K + k] = v;
        s += (double)v * (double)v;
    }
    c2[k] = (float)s;
}

struct Best {
    float d0; int i0; float d1;     // best distance, its index, second-best distance
};
__device__ __forceinline__ void best_push(Best& b, float d, int i) {
    if (d < b.d0 || (d == b.d0 && i < b.i0)) { b.d1 = b.d0; b.d0 = d; b.i0 = i; }
    else if (d < b.d1) b.d1 = d;
}
__device__ __forceinline__ Best best_wave_reduce(Best b) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float od0 = __shfl_xor(b.d0, o, 64), od1 = __shfl_xor(b.d1, o, 64);
        const int oi0 = __shfl_xor(b.i0, o, 64);
        if (od0 < b.d0 || (od0 == b.d0 && oi0 < b.i0)) { b.d1 = fminf(b.d0, od1); b.d0 = od0; b.i0 = oi0; }
        else b.d1 = fminf(b.d1, od0);
    }
    return b;
}

// 16 rows of z per workgroup; every thread owns KPT codes (k = tid + 256*kk) for all 16 rows.
// Dot products are accumulated in fp64 (FP64 vector rate on CDNA4 is half the fp32 rate; this kernel is
// tiny) and rounded once, so the only fp32 roundings left are the ones the reference formula itself has.
template <int KPT>
__global__ __launch_bounds__(256) void vq_nearest_kernel(const float* __restrict__ z, const float* __restrict__ cbt,
                                                         const float* __restrict__ c2, long M, int D, int K,
                                                         int64_t* __restrict__ idx, float* __restrict__ margin) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* zs = (float*)smem_raw;                    // [16][D]
    float* x2s = zs + 16 * D;                        // [16]
    float* ds = x2s + 16;                            // [16][K]
    const long r0 = (long)blockIdx.x * 16;
    for (int e = threadIdx.x; e < 16 * D / 4; e += 256) {
        const int r = e / (D / 4), c = (e - r * (D / 4)) * 4;
        *(f32x4*)(zs + r * D + c) = (r0 + r < M) ? *(const f32x4*)(z + (r0 + r) * D + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        double s = 0.0;
        for (int d = 0; d < D; ++d) s += (double)zs[threadIdx.x * D + d] * (double)zs[threadIdx.x * D + d];
        x2s[threadIdx.x] = (float)s;
    }
    double acc[KPT][16];
#pragma unroll
    for (int kk = 0; kk < KPT; ++kk)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kk][r] = 0.0;
    for (int d = 0; d < D; d += 4) {
        float c[KPT][4];
#pragma unroll
        for (int kk = 0; kk < KPT; ++kk) {
            const int k = threadIdx.x + 256 * kk;
#pragma unroll
            for (int e = 0; e < 4; ++e) c[kk][e] = (k < K) ? cbt[(long)(d + e) * K + k] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x4 zv = *(const f32x4*)(zs + r * D + d);
#pragma unroll
            for (int kk = 0; kk < KPT; ++kk)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[kk][r] = fma((double)zv[e], (double)c[kk][e], acc[kk][r]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KPT; ++kk) {
        const int k = threadIdx.x + 256 * kk;
        if (k < K) {
            const float ck = c2[k];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float s = ck + x2s[r];                       // fl32(|c|^2 + |z|^2)
                ds[r * K + k] = fmaf(-2.0f, (float)acc[kk][r], s); // fl32(s - 2*dot)
            }
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < 16; r += 4) {
        if (r0 + r >= M) break;
        Best b{INFINITY, 0x7fffffff, INFINITY};
        for (int k = lane; k < K; k += 64) best_push(b, ds[r * K + k], k);
        b = best_wave_reduce(b);
        if (lane == 0) {
            idx[r0 + r] = b.i0;
            if (margin) margin[r0 + r] = b.d1 - b.d0;
        }
    }
}

// The same quantiser on the fp64 matrix cores (v_mfma_f64_16x16x4_f64: D[m][n] += sum_k A[m][k] B[n][k], A lane (m = lane & 15,
// g = lane >> 4) feeds A[m][k = g], B lane (n, g) feeds B[n][g], result lane (n, g) holds D[g + 4e][n], e = 0..3 -- the fp64
// variant interleaves the rows, unlike the 32-bit MFMAs' D[4g + e][n]).  A = 16 rows of z
// (straight from global: a lane re-reads its row 4 bytes at a time, L1-resident), B = 16 codes of the transposed codebook (64-byte
// segments per k), so a lane ends up with the fp64 dot products of ITS code with four rows; the distance keeps the reference formula
// fl32(fl32(|c|^2 + |z|^2) - 2 <z, c>) and the first minimum wins.  Workgroup = RT x 16 rows; its NWV waves split the codebook
// (NT tiles of 16 codes each); the per-row (best, index, second best) triples are merged across the 16 lanes of a code group, then
// across the waves through LDS.  The dot products differ from the VALU kernel's only in fp64 summation order (1e-16 relative).
// The thread-per-code kernel above converted every operand to fp64 per use: 5.9 ms for 40 960 vectors at D = 1024, K = 512.
typedef __attribute__((ext_vector_type(4))) double f64x4;

__device__ __forceinline__ void best_merge(Best& b, float od0, int oi0, float od1) {
    if (od0 < b.d0 || (od0 == b.d0 && oi0 < b.i0)) { b.d1 = fminf(b.d0, od1); b.d0 = od0; b.i0 = oi0; }
    else b.d1 = fminf(b.d1, od0);
}

template <int NT, int RT, int NWV>
__global__ __launch_bounds__(64 * NWV) void vq_nearest_mfma_kernel(const float* __restrict__ z, const float* __restrict__ cbt,
                                                              const float* __restrict__ c2, long M, int D, int K,
                                                              int64_t* __restrict__ idx, float* __restrict__ margin) {
    __shared__ float sd0[NWV][RT * 16], sd1[NWV][RT * 16];
    __shared__ int si0[NWV][RT * 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l15 = lane & 15, g = lane >> 4;
    const long r0 = (long)blockIdx.x * (RT * 16);
    const int code0 = wave * NT * 16;                                  // this wave's slice of the codebook
    const float* zrow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) zrow[rt] = z + min(r0 + rt * 16 + l15, M - 1) * (long)D + g;
    int ccol[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) ccol[t] = min(code0 + t * 16 + l15, K - 1);
    f64x4 acc[RT][NT];
    double x2[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        x2[rt] = 0.0;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[rt][t] = f64x4{0.0, 0.0, 0.0, 0.0};
    }
    for (int d = 0; d < D; d += 4) {
        double a[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            a[rt] = (double)zrow[rt][d];
            x2[rt] = fma(a[rt], a[rt], x2[rt]);
        }
        const float* cb = cbt + (long)(d + g) * K;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const double b = (double)cb[ccol[t]];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt], b, acc[rt][t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        // |z|^2 of row l15: this lane summed its k = g slice
        double s = x2[rt];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float x2f = (float)s;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xr = __shfl(x2f, g + 4 * e);                   // |z|^2 of the row this accumulator element belongs to
            Best b{INFINITY, 0x7fffffff, INFINITY};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int code = code0 + t * 16 + l15;
                if (code < K) {
                    const float sck = c2[code] + xr;                                    // fl32(|c|^2 + |z|^2)
                    best_push(b, fmaf(-2.0f, (float)acc[rt][t][e], sck), code);         // fl32(s - 2*dot)
                }
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {                          // across the 16 lanes (codes) of this lane group
                const float od0 = __shfl_xor(b.d0, o), od1 = __shfl_xor(b.d1, o);
                const int oi0 = __shfl_xor(b.i0, o);
                best_merge(b, od0, oi0, od1);
            }
            if (l15 == 0) {
                const int r = rt * 16 + g + 4 * e;
                sd0[wave][r] = b.d0;
                si0[wave][r] = b.i0;
                sd1[wave][r] = b.d1;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < RT * 16 && r0 + threadIdx.x < M) {
        Best b{sd0[0][threadIdx.x], si0[0][threadIdx.x], sd1[0][threadIdx.x]};
#pragma unroll
        for (int w = 1; w < NWV; ++w) best_merge(b, sd0[w][threadIdx.x], si0[w][threadIdx.x], sd1[w][threadIdx.x]);
        idx[r0 + threadIdx.x] = b.i0;
        if (margin) margin[r0 + threadIdx.x] = b.d1 - b.d0;
    }
}

template <typename OT>
__global__ __launch_bounds__(256) void embedding_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                        OT* __restrict__ out, long n, int C, int n_table, int relu,
                                                        long group, long group_stride, long off, long inner, long inner_stride,
                                                        int* __restrict__ err) {
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    long id = ids[i];
    if (id < 0 || id >= n_table) {          // the reference raises IndexError here: reported by mage_check_device_errors
        if (lane == 0) mage_raise(err, MAGE_DEVERR_EMBEDDING_ID, id, n_table);
        id = id < 0 ? 0 : n_table - 1;      // stay memory-safe meanwhile
    }
    const long ig = i % group;
    const long orow = (i / group) * group_stride + (ig / inner) * inner_stride + (ig % inner) + off;
    const float* src = table + id * C;
    OT* dst = out + orow * C;
    for (int c = lane * 4; c < C; c += 256) {
        f32x4 v = *(const f32x4*)(src + c);
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        store4(dst + c, v);
    }
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, long rows, int K, long ld, long group,
                                                     long in_stride, long in_off, int64_t* __restrict__ out,
                                                     long out_stride, long out_off, float* __restrict__ margin) {
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    const int lane = threadIdx.x & 63;
    const long gi = i / group, gr = i - gi * group;
    const float* p = logits + (gi * in_stride + gr + in_off) * ld;
    Best b{INFINITY, 0x7fffffff, INFINITY};          // minimise the negated logit: first maximum wins
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 v = *(const f32x4*)(p + k);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k + e < K) best_push(b, -v[e], k + e);
    }
    b = best_wave_reduce(b);
    if (lane == 0) {
        out[gi * out_stride + gr + out_off] = b.i0;
        if (margin) margin[i] = b.d1 - b.d0;
    }
}

__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                      long rows, int K, float* __restrict__ row_loss, int* __restrict__ err) {
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = logits + i * K;
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, p[k]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += expf(p[k] - mx);
    s = wave_sum(s);
    if (lane == 0) {
        long tg = target[i];
        if (tg < 0 || tg >= K) {
            mage_raise(err, MAGE_DEVERR_CE_TARGET, tg, K);
            tg = 0;
        }
        row_loss[i] = (logf(s) + mx) - p[tg];
    }
}

__global__ __launch_bounds__(1024) void sum_kernel(const float* __restrict__ v, long n, float* __restrict__ out, double inv) {
    __shared__ double red[16];
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += 1024) s += (double)v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        out[0] = (float)(t * inv);
    }
}

}  // namespace

extern "C" int mage_vq_prepare(const float* codebook, int32_t K, int32_t D, float* codebook_t, float* c2, void* stream) {
    MAGE_CHECK_ARG(codebook && codebook_t && c2 && K > 0 && D > 0, "mage_vq_prepare: bad arguments");
    hipLaunchKernelGGL(vq_prepare_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, codebook, K, D,
                       codebook_t, c2);
    MAGE_CHECK_LAUNCH("mage_vq_prepare");
    return MAGE_OK;
}

extern "C" int mage_vq_nearest(const float* z, const float* codebook_t, const float* c2, int64_t M, int32_t D, int32_t K,
                               int64_t* idx, float* margin, void* stream) {
    MAGE_CHECK_ARG(z && codebook_t && c2 && idx, "mage_vq_nearest: null pointer");
    MAGE_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && K > 0 && K <= 1024, "mage_vq_nearest: M=%ld D=%d K=%d unsupported", (long)M, D, K);
    hipStream_t s = (hipStream_t)stream;
    if (!mage_options().vq_no_mfma) {                                       // the fp64 matrix-core kernel: any D % 4 == 0, K <= 1024
        const dim3 g32((unsigned)((M + 31) / 32));        // 4 tiles of 16 codes per wave: 92 + 64 registers, 3 waves per SIMD
        if (K <= 256) hipLaunchKernelGGL((vq_nearest_mfma_kernel<4, 2, 4>), g32, dim3(256), 0, s, z, codebook_t, c2, (long)M, D, K, idx, margin);
        else if (K <= 512) hipLaunchKernelGGL((vq_nearest_mfma_kernel<4, 2, 8>), g32, dim3(512), 0, s, z, codebook_t, c2, (long)M, D, K, idx, margin);
        else hipLaunchKernelGGL((vq_nearest_mfma_kernel<8, 2, 8>), g32, dim3(512), 0, s, z, codebook_t, c2, (long)M, D, K, idx, margin);
        MAGE_CHECK_LAUNCH("mage_vq_nearest");
        return MAGE_OK;
    }
    const size_t lds = (size_t)(16 * D + 16 + 16 * K) * 4;
    MAGE_CHECK_ARG(lds <= 144 * 1024, "mage_vq_nearest: D=%d K=%d exceed the LDS budget", D, K);
    static bool attr_set[MAGE_MAX_DEVICES] = {false};      // the attribute is per device (idempotent: a racing second call is harmless)
    const int dev = mage_device_index();
    MAGE_CHECK_ARG(dev >= 0, "mage_vq_nearest: no current device");
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)vq_nearest_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        (void)hipFuncSetAttribute((const void*)vq_nearest_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        (void)hipFuncSetAttribute((const void*)vq_nearest_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        attr_set[dev] = true;
    }
    const dim3 grid((unsigned)((M + 15) / 16)), blk(256);
    if (K <= 256) hipLaunchKernelGGL((vq_nearest_kernel<1>), grid, blk, lds, s, z, codebook_t, c2, (long)M, D, K, idx, margin);
    else if (K <= 512) hipLaunchKernelGGL((vq_nearest_kernel<2>), grid, blk, lds, s, z, codebook_t, c2, (long)M, D, K, idx, margin);
    else hipLaunchKernelGGL((vq_nearest_kernel<4>), grid, blk, lds, s, z, codebook_t, c2, (long)M, D, K, idx, margin);
    MAGE_CHECK_LAUNCH("mage_vq_nearest");
    return MAGE_OK;
}

// ------------------------------------------------------------------------------------ convolution of an embedding as a table sum
// A k x k convolution (stride 1, zero padding (k-1)/2) over nn.Embedding rows has only n_codes distinct input vectors, so it is
//     y[img, p, :] = pos[p, :] + sum over taps (ky, kx) of T[tap][ids[img, p + (ky, kx) - centre], :]      (taps outside the image: nothing)
// with T[tap][code] = W_tap emb[code] precomputed once per weights (and any Linear applied to the result folded into T and pos).
// One wave per output pixel: taps_h*taps_w rows of C floats gathered from a table that lives in L2 / Infinity Cache (9 x 512 x 512
// fp32 = 9.4 MB at the MNIST config), summed in a fixed order in fp32, + a broadcast row table (the T positions), written to row
// yrow = (m / group) * y_group_stride + m % group + y_off of y (m = img*H*W + p): the frame slots of the decoder's residual stream.
namespace {
template <typename TT_, typename OT, int VPL>
__global__ __launch_bounds__(256) void table_conv_kernel(const int64_t* __restrict__ ids, const TT_* __restrict__ table, const float* __restrict__ pos,
                                                         const float* __restrict__ bias, int relu,
                                                         const float* __restrict__ rowadd, OT* __restrict__ y, long n_pix, int H, int W,
                                                         int th, int tw, int n_codes, int C, long group, long y_group_stride, long y_off,
                                                         long rowadd_div, int rowadd_mod, long ldy, int* __restrict__ err) {
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= n_pix) return;
    const int lane = threadIdx.x & 63;
    const int plane = H * W;
    const long img = m / plane;
    const int p = (int)(m - img * plane);
    const int py = p / W, px = p - py * W;
    f32x4 acc[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int c = j * 256 + lane * 4;
        acc[j] = (pos && c < C) ? *(const f32x4*)(pos + (long)p * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (bias && c < C) acc[j] += *(const f32x4*)(bias + c);
    }
    const int64_t* img_ids = ids + img * plane;
    for (int ky = 0; ky < th; ++ky) {
        const int iy = py + ky - (th >> 1);
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int kx = 0; kx < tw; ++kx) {
            const int ix = px + kx - (tw >> 1);
            if ((unsigned)ix >= (unsigned)W) continue;
            long id = img_ids[iy * W + ix];
            if (id < 0 || id >= n_codes) {      // the reference's nn.Embedding raises IndexError: reported by mage_check_device_errors
                if (lane == 0) mage_raise(err, MAGE_DEVERR_EMBEDDING_ID, id, n_codes);
                id = id < 0 ? 0 : n_codes - 1;
            }
            MAGE_DASSERT(id >= 0 && id < n_codes);
            const TT_* row = table + ((long)(ky * tw + kx) * n_codes + id) * C;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const int c = j * 256 + lane * 4;
                if (c < C) acc[j] += load4(row + c);
            }
        }
    }
    const long yrow = (m / group) * y_group_stride + m % group + y_off;
    if (rowadd) {
        const float* rp = rowadd + ((yrow / rowadd_div) % rowadd_mod) * (long)C;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int c = j * 256 + lane * 4;
            if (c < C) acc[j] += *(const f32x4*)(rp + c);
        }
    }
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int c = j * 256 + lane * 4;
        if (c < C) {
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][e] = fmaxf(acc[j][e], 0.f);
            }
            store4(y + yrow * ldy + c, acc[j]);
        }
    }
}
// The same sums for 16-bit tables and rows of exactly 512 channels (the decoder's frame fill: 245 760 pixels per call at cfg2): a lane owns 8
// channels, so a table row is ONE 16-byte load per lane (the kernel above takes two 8-byte loads per row: twice the load instructions at
// 0.54-0.70x the bytes per instruction).  Same order of additions per element: pos (+ bias), the taps in (ky, kx) order, the row table.
template <typename T16>
__global__ __launch_bounds__(256) void table_conv512_kernel(const int64_t* __restrict__ ids, const T16* __restrict__ table, const float* __restrict__ pos,
                                                            const float* __restrict__ bias, int relu, const float* __restrict__ rowadd,
                                                            T16* __restrict__ y, long n_pix, int H, int W, int th, int tw, int n_codes, long group,
                                                            long y_group_stride, long y_off, long rowadd_div, int rowadd_mod, long ldy,
                                                            int* __restrict__ err) {
    constexpr int C = 512;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= n_pix) return;
    const int lane = threadIdx.x & 63, c = lane * 8;
    const int plane = H * W;
    const long img = m / plane;
    const int p = (int)(m - img * plane);
    const int py = p / W, px = p - py * W;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    if (pos) {
        a0 = *(const f32x4*)(pos + (long)p * C + c);
        a1 = *(const f32x4*)(pos + (long)p * C + c + 4);
    }
    if (bias) {
        a0 += *(const f32x4*)(bias + c);
        a1 += *(const f32x4*)(bias + c + 4);
    }
    const int64_t* img_ids = ids + img * plane;
    for (int ky = 0; ky < th; ++ky) {
        const int iy = py + ky - (th >> 1);
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int kx = 0; kx < tw; ++kx) {
            const int ix = px + kx - (tw >> 1);
            if ((unsigned)ix >= (unsigned)W) continue;
            long id = img_ids[iy * W + ix];
            if (id < 0 || id >= n_codes) {
                if (lane == 0) mage_raise(err, MAGE_DEVERR_EMBEDDING_ID, id, n_codes);
                id = id < 0 ? 0 : n_codes - 1;
            }
            const u32x4 r = *(const u32x4*)(table + ((long)(ky * tw + kx) * n_codes + id) * C + c);
            a0 += widen4<T16>(uint2{r[0], r[1]});
            a1 += widen4<T16>(uint2{r[2], r[3]});
        }
    }
    const long yrow = (m / group) * y_group_stride + m % group + y_off;
    if (rowadd) {
        const float* rp = rowadd + ((yrow / rowadd_div) % rowadd_mod) * (long)C;
        a0 += *(const f32x4*)(rp + c);
        a1 += *(const f32x4*)(rp + c + 4);
    }
    if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a0[e] = fmaxf(a0[e], 0.f);
            a1[e] = fmaxf(a1[e], 0.f);
        }
    }
    store8(y + yrow * ldy + c, a0, a1);
}
}  // namespace

extern "C" int mage_table_conv(const int64_t* ids, int64_t n_img, int32_t H, int32_t W, int32_t taps_h, int32_t taps_w, const void* table,
                               int32_t table_dtype, int32_t n_codes, int32_t C, const float* pos, const float* bias, int32_t relu,
                               const float* rowadd, int64_t rowadd_div, int32_t rowadd_mod, void* y, int32_t y_dtype, int64_t ldy, int64_t group,
                               int64_t y_group_stride, int64_t y_off, void* stream) {
    MAGE_CHECK_ARG(ids && table && y, "mage_table_conv: null pointer");
    MAGE_CHECK_ARG(n_img > 0 && H > 0 && W > 0 && taps_h >= 1 && taps_w >= 1 && (taps_h & 1) && (taps_w & 1) && n_codes > 0 && C > 0 && C % 4 == 0 &&
                       C <= 2048 && ldy >= C && ldy % 4 == 0 && group > 0,
                   "mage_table_conv: bad sizes (odd taps, C %% 4 == 0, C <= 2048)");
    MAGE_CHECK_ARG(!rowadd || (rowadd_div >= 1 && rowadd_mod >= 1), "mage_table_conv: bad rowadd div/mod");
    const bool ysplit = y_dtype == MAGE_BF16X3 || y_dtype == MAGE_F16X3;
    MAGE_CHECK_ARG((table_dtype == MAGE_F32 || table_dtype == MAGE_BF16 || table_dtype == MAGE_F16) &&
                       (y_dtype == MAGE_F32 || y_dtype == MAGE_BF16 || y_dtype == MAGE_F16 || ysplit) &&
                       (table_dtype == MAGE_F32 || y_dtype == MAGE_F32 || table_dtype == y_dtype),
                   "mage_table_conv: bad table / y dtype %d %d (fp32, bf16 or f16; a 16-bit table writes fp32 or its own type)", table_dtype, y_dtype);
    MAGE_CHECK_ARG(!ysplit || (C % 64 == 0 && ldy == C && (((uintptr_t)y) & 255) == 0 && table_dtype == MAGE_F32),
                   "mage_table_conv: split output needs an fp32 table, C %% 64 == 0, packed rows, y 256-byte aligned");
    int* err = mage_error_word();
    MAGE_CHECK_ARG(err != nullptr, "mage_table_conv: mage_init() has not been called");
    const long n_pix = (long)n_img * H * W;
    const dim3 grid((unsigned)((n_pix + 3) / 4)), blk(256);
    hipStream_t s = (hipStream_t)stream;
#define TC(T_, O_, V) hipLaunchKernelGGL((table_conv_kernel<T_, O_, V>), grid, blk, 0, s, ids, (const T_*)table, pos, bias, relu, rowadd, (O_*)y, n_pix, H, W, \
                                         taps_h, taps_w, n_codes, C, (long)group, (long)y_group_stride, (long)y_off, (long)rowadd_div, rowadd_mod, (long)ldy, err)
#define TCV(T_, O_) do { if (vpl <= 1) TC(T_, O_, 1); else if (vpl <= 2) TC(T_, O_, 2); else if (vpl <= 4) TC(T_, O_, 4); else TC(T_, O_, 8); } while (0)
    const int vpl = (C + 255) / 256;
    if (C == 512 && table_dtype == y_dtype && (y_dtype == MAGE_BF16 || y_dtype == MAGE_F16) && ldy % 8 == 0 &&
        (((uintptr_t)table | (uintptr_t)y | (uintptr_t)pos | (uintptr_t)bias | (uintptr_t)rowadd) & 15) == 0) {
        if (y_dtype == MAGE_BF16)
            hipLaunchKernelGGL((table_conv512_kernel<unsigned short>), grid, blk, 0, s, ids, (const unsigned short*)table, pos, bias, relu, rowadd,
                               (unsigned short*)y, n_pix, H, W, taps_h, taps_w, n_codes, (long)group, (long)y_group_stride, (long)y_off, (long)rowadd_div,
                               rowadd_mod, (long)ldy, err);
        else
            hipLaunchKernelGGL((table_conv512_kernel<f16_t>), grid, blk, 0, s, ids, (const f16_t*)table, pos, bias, relu, rowadd, (f16_t*)y, n_pix, H, W,
                               taps_h, taps_w, n_codes, (long)group, (long)y_group_stride, (long)y_off, (long)rowadd_div, rowadd_mod, (long)ldy, err);
        MAGE_CHECK_LAUNCH("mage_table_conv");
        return MAGE_OK;
    }
    if (y_dtype == MAGE_F16X3) TCV(float, split_f16);
    else if (y_dtype == MAGE_BF16X3) TCV(float, split_bf16);
    else if (table_dtype == MAGE_F32 && y_dtype == MAGE_F32) TCV(float, float);
    else if (table_dtype == MAGE_F32 && y_dtype == MAGE_F16) TCV(float, f16_t);
    else if (table_dtype == MAGE_F16 && y_dtype == MAGE_F32) TCV(f16_t, float);
    else if (table_dtype == MAGE_F16) TCV(f16_t, f16_t);
    else if (table_dtype == MAGE_F32) TCV(float, unsigned short);
    else if (y_dtype == MAGE_F32) TCV(unsigned short, float);
    else TCV(unsigned short, unsigned short);
#undef TCV
#undef TC
    MAGE_CHECK_LAUNCH("mage_table_conv");
    return MAGE_OK;
}

extern "C" int mage_embedding(const int64_t* ids, const float* table, void* out, int32_t out_dtype, int64_t n, int32_t C,
                              int32_t n_table, int32_t relu, int64_t group, int64_t group_stride, int64_t off, int64_t inner,
                              int64_t inner_stride, void* stream) {
    MAGE_CHECK_ARG(ids && table && out, "mage_embedding: null pointer");
    MAGE_CHECK_ARG(n > 0 && C > 0 && C % 4 == 0 && n_table > 0 && group > 0, "mage_embedding: bad sizes n=%ld C=%d", (long)n, C);
    if (inner <= 0) {                      // one-level grouping
        inner = group;
        inner_stride = group;
    }
    const dim3 grid((unsigned)((n + 3) / 4)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    int* err = mage_error_word();
    MAGE_CHECK_ARG(err != nullptr, "mage_embedding: mage_init() has not been called");
    if (out_dtype == MAGE_F32)
        hipLaunchKernelGGL((embedding_kernel<float>), grid, blk, 0, s, ids, table, (float*)out, (long)n, C, n_table, relu,
                           (long)group, (long)group_stride, (long)off, (long)inner, (long)inner_stride, err);
    else if (out_dtype == MAGE_BF16)
        hipLaunchKernelGGL((embedding_kernel<unsigned short>), grid, blk, 0, s, ids, table, (unsigned short*)out, (long)n, C,
                           n_table, relu, (long)group, (long)group_stride, (long)off, (long)inner, (long)inner_stride, err);
    else if (out_dtype == MAGE_F16)
        hipLaunchKernelGGL((embedding_kernel<f16_t>), grid, blk, 0, s, ids, table, (f16_t*)out, (long)n, C, n_table, relu, (long)group,
                           (long)group_stride, (long)off, (long)inner, (long)inner_stride, err);
    else if (out_dtype == MAGE_BF16X3 || out_dtype == MAGE_F16X3) {     // split rows (common.h): the frame convolution's A operand in the fast parity mode
        MAGE_CHECK_ARG(C % 64 == 0 && (((uintptr_t)out) & 255) == 0, "mage_embedding: split output needs C %% 64 == 0 and out 256-byte aligned");
        if (out_dtype == MAGE_BF16X3)
            hipLaunchKernelGGL((embedding_kernel<split_bf16>), grid, blk, 0, s, ids, table, (split_bf16*)out, (long)n, C, n_table, relu,
                               (long)group, (long)group_stride, (long)off, (long)inner, (long)inner_stride, err);
        else
            hipLaunchKernelGGL((embedding_kernel<split_f16>), grid, blk, 0, s, ids, table, (split_f16*)out, (long)n, C, n_table, relu,
                               (long)group, (long)group_stride, (long)off, (long)inner, (long)inner_stride, err);
    } else {
        mage_set_error("mage_embedding: bad out_dtype %d", out_dtype);
        return MAGE_EINVAL;
    }
    MAGE_CHECK_LAUNCH("mage_embedding");
    return MAGE_OK;
}

extern "C" int mage_argmax(const float* logits, int64_t rows, int32_t K, int64_t ld, int64_t group, int64_t in_group_stride,
                           int64_t in_off, int64_t* out, int64_t out_group_stride, int64_t out_off, float* margin,
                           void* stream) {
    MAGE_CHECK_ARG(logits && out, "mage_argmax: null pointer");
    MAGE_CHECK_ARG(rows > 0 && K > 0 && K % 4 == 0 && ld % 4 == 0 && group > 0, "mage_argmax: bad sizes rows=%ld K=%d", (long)rows, K);
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, (long)rows,
                       K, (long)ld, (long)group, (long)in_group_stride, (long)in_off, out, (long)out_group_stride,
                       (long)out_off, margin);
    MAGE_CHECK_LAUNCH("mage_argmax");
    return MAGE_OK;
}

extern "C" int mage_cross_entropy(const float* logits, const int64_t* target, int64_t rows, int32_t K, float* row_loss,
                                  float* loss_mean, void* stream) {
    MAGE_CHECK_ARG(logits && target && row_loss && loss_mean, "mage_cross_entropy: null pointer");
    MAGE_CHECK_ARG(rows > 0 && K > 0, "mage_cross_entropy: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    int* err = mage_error_word();
    MAGE_CHECK_ARG(err != nullptr, "mage_cross_entropy: mage_init() has not been called");
    hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, target, (long)rows, K, row_loss, err);
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, s, row_loss, (long)rows, loss_mean, 1.0 / (double)rows);
    MAGE_CHECK_LAUNCH("mage_cross_entropy");
    return MAGE_OK;
}
