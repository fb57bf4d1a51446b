// Training path: the kernels of MAGE.forward's backward pass that are not GEMMs.
//
// The dense gradients reuse mage_gemm: dX = dY W is the forward kernel on a transposed weight copy; dW = dY^T X contracts over
// ALL M tokens while its output is only [N, K], so both operands are first transposed (mage_transpose: HBM-bound, one pass) and
// the product runs as one split-K launch of the same MFMA kernel (mage_gemm_desc::n_split) whose partial outputs
// mage_sum_partials adds in a fixed order.  Everything here is deterministic except mage_embedding_bwd (fp32 atomics, like the
// reference's own nn.Embedding backward on a GPU).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ transpose (+ conv tap gather)
// y[(c + y_row0) * ldy + m] = x[arow(m) * ldx + c]   for m < M,  0 for M <= m < Mp  (the split-K GEMM reads whole 64-column
// slabs, so the tail up to the padded width is zero-filled).  arow(m) is the implicit-GEMM row map of mage_gemm: m -> (img, oy,
// ox) over an out_h x out_w plane, row = img*img_stride + (oy+dy)*in_w + (ox+dx) + a_off, zero outside [0,in_h) x [0,in_w): a
// plain transpose is out_h = 1, out_w = M; the nine shifted copies of the activation that the conv3x3 weight gradient
// contracts with are nine calls with (dy, dx) = tap - 1 and y_row0 = tap * C.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, long y_row0,
                                                        long M, long Mp, int C, int out_h, int out_w, int in_h, int in_w,
                                                        long img_stride, long a_off, int dy, int dx, int stride) {
    __shared__ T tile[64][65];
    const long m0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 x 4
    const long plane = (long)out_h * out_w;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {                           // r: m within the tile; tx: channel
        const long m = m0 + r;
        T v = (T)0;
        if (m < M && c0 + tx < C) {
            const long img = m / plane, rem = m - img * plane;
            const int oy = (int)(rem / out_w), ox = (int)(rem - (long)oy * out_w);
            const int iy = oy * stride + dy, ix = ox * stride + dx;
            if ((unsigned)iy < (unsigned)in_h && (unsigned)ix < (unsigned)in_w)
                v = x[(img * img_stride + (long)iy * in_w + ix + a_off) * ldx + c0 + tx];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {                           // r: channel within the tile; tx: m
        const long m = m0 + tx;
        if (c0 + r < C && m < Mp) y[(y_row0 + c0 + r) * ldy + m] = tile[tx][r];
    }
}


// 16-byte variant for bf16 with C % 8 == 0: a thread loads 8 consecutive channels of one (gathered) row, the 64 x 64 tile sits in LDS
// row-major (pitch 132 B: the column reads below touch distinct banks), then every thread assembles 8 consecutive m of one channel
// and stores 16 bytes: both sides of the transposition move whole 128-byte segments (the 2-byte version ran at 1.1 TB/s).
__global__ __launch_bounds__(256) void transpose8_kernel(const unsigned short* __restrict__ x, long ldx, unsigned short* __restrict__ y,
                                                         long ldy, long y_row0, long M, long Mp, int C, int out_h, int out_w, int in_h,
                                                         int in_w, long img_stride, long a_off, int dy, int dx, int stride) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64 * 66];
    const long m0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const long plane = (long)out_h * out_w;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = threadIdx.x + 256 * k;          // 512 (row, 8-channel chunk) pairs
        const int r = e >> 3, cc = (e & 7) * 8;
        const long m = m0 + r;
        uint4 v = uint4{0u, 0u, 0u, 0u};
        if (m < M && c0 + cc < C) {
            const long img = m / plane, rem = m - img * plane;
            const int oy = (int)(rem / out_w), ox = (int)(rem - (long)oy * out_w);
            const int iy = oy * stride + dy, ix = ox * stride + dx;
            if ((unsigned)iy < (unsigned)in_h && (unsigned)ix < (unsigned)in_w)
                v = *(const uint4*)(x + (img * img_stride + (long)iy * in_w + ix + a_off) * ldx + c0 + cc);
        }
        unsigned* t32 = (unsigned*)(tile + r * 66 + cc);      // 66-element pitch: 4-byte aligned
        t32[0] = v.x; t32[1] = v.y; t32[2] = v.z; t32[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = threadIdx.x + 256 * k;          // 512 (channel, 8-row chunk) pairs
        const int c = e >> 3, mm = (e & 7) * 8;
        if (c0 + c < C && m0 + mm < Mp) {
            unsigned short h[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = tile[(mm + j) * 66 + c];
            uint4 o;
            o.x = h[0] | ((unsigned)h[1] << 16); o.y = h[2] | ((unsigned)h[3] << 16);
            o.z = h[4] | ((unsigned)h[5] << 16); o.w = h[6] | ((unsigned)h[7] << 16);
            *(uint4*)(y + (y_row0 + c0 + c) * ldy + m0 + mm) = o;
        }
    }
}

// The same transposition with the column sums of x folded in (the bias gradient db = sum_m dY[m, :] of a Linear, which used to be a
// second pass over the transposed copy: 7 ms of a 119 ms training step): a workgroup walks TPW consecutive 64-row tiles, every thread adds
// up the 8 rows it assembles per channel, the 8 threads of a channel combine by xor-shuffles, and the workgroup's sums land in
// colsum[blockIdx.x][c] (fixed order; mage_sum_partials adds the ceil(tiles / TPW) partial rows).
constexpr int TRANSPOSE_TPW = 16;
__global__ __launch_bounds__(256) void transpose8_colsum_kernel(const unsigned short* __restrict__ x, long ldx, unsigned short* __restrict__ y,
                                                                long ldy, long M, long Mp, int C, int out_h, int out_w, int in_h, int in_w,
                                                                long img_stride, long a_off, float* __restrict__ colsum) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64 * 66];
    const int c0 = blockIdx.y * 64;
    const long plane = (long)out_h * out_w;
    float cs[2] = {0.f, 0.f};
    for (int t = 0; t < TRANSPOSE_TPW; ++t) {
        const long m0 = ((long)blockIdx.x * TRANSPOSE_TPW + t) * 64;
        if (m0 >= Mp) break;                                              // block-uniform
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = threadIdx.x + 256 * k;
            const int r = e >> 3, cc = (e & 7) * 8;
            const long m = m0 + r;
            uint4 v = uint4{0u, 0u, 0u, 0u};
            if (m < M && c0 + cc < C) {
                const long img = m / plane, rem = m - img * plane;
                const int oy = (int)(rem / out_w), ox = (int)(rem - (long)oy * out_w);
                if ((unsigned)oy < (unsigned)in_h && (unsigned)ox < (unsigned)in_w)
                    v = *(const uint4*)(x + (img * img_stride + (long)oy * in_w + ox + a_off) * ldx + c0 + cc);
            }
            unsigned* t32 = (unsigned*)(tile + r * 66 + cc);
            t32[0] = v.x; t32[1] = v.y; t32[2] = v.z; t32[3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = threadIdx.x + 256 * k;
            const int c = e >> 3, mm = (e & 7) * 8;
            if (c0 + c < C && m0 + mm < Mp) {
                unsigned short h[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    h[j] = tile[(mm + j) * 66 + c];
                    cs[k] += bf_bits2f(h[j]);
                }
                uint4 o;
                o.x = h[0] | ((unsigned)h[1] << 16); o.y = h[2] | ((unsigned)h[3] << 16);
                o.z = h[4] | ((unsigned)h[5] << 16); o.w = h[6] | ((unsigned)h[7] << 16);
                *(uint4*)(y + (long)(c0 + c) * ldy + m0 + mm) = o;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float v = cs[k];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        const int e = threadIdx.x + 256 * k;
        const int c = e >> 3;
        if ((e & 7) == 0 && c0 + c < C) colsum[(long)blockIdx.x * C + c0 + c] = v;
    }
}

// out[r] = sum_c x[r*ld + c], c < n (fp32 accumulation, fixed order): bias gradients from the transposed dY.
// blockIdx.y = column chunk (a row of 10^5 columns on ONE wave was 0.76 ms per call: 19 % of a training step): partial sums
// out[chunk][r] for mage_sum_partials.
template <typename T>
__global__ __launch_bounds__(256) void row_sum_kernel(const T* __restrict__ x, long ld, long n, int rows, float* __restrict__ out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const long per = ((n + gridDim.y - 1) / gridDim.y + 7) / 8 * 8;
    const long c0 = (long)blockIdx.y * per, c1 = min(n, c0 + per);
    const T* p = x + (long)r * ld;
    float s = 0.f;
    for (long c = c0 + lane; c < c1; c += 64) s += to_f32<T>(p[c]);
    s = wave_sum(s);
    if (lane == 0) out[(long)blockIdx.y * rows + r] = s;
}

// out[i] = (accumulate ? out[i] : 0) + sum_s part[s*stride + i]
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, long stride, int n_part, long n,
                                                           float* __restrict__ out, int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? out[i] : 0.f;
    for (int p = 0; p < n_part; ++p) s += part[(long)p * stride + i];
    out[i] = s;
}

// The same sum for 16-byte aligned operands: a workgroup owns 64 float4 columns, its four waves take the partials p = w, w+4, w+8, ..
// (independent 16-byte loads, 4 in flight per lane) and their four sums are added in a fixed order through LDS:
//     out = ((s_0 + s_1) + (s_2 + s_3)) [+ out],  s_w = part[w] + part[w+4] + ...      -- deterministic, like the scalar kernel's order.
// The scalar kernel walked 64 dependent 4-byte loads per thread: 6.9 ms per training step at B = 64 for ~2.7 GB of split-K partials.
__global__ __launch_bounds__(256) void sum_partials4_kernel(const float* __restrict__ part, long stride, int n_part, long n4,
                                                            float* __restrict__ out, int accumulate) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long i4 = (long)blockIdx.x * 64 + lane;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i4 < n4) {
        const float* p0 = part + i4 * 4;
        int p = w;
        for (; p + 12 < n_part; p += 16) {
            const f32x4 a = *(const f32x4*)(p0 + (long)p * stride), b = *(const f32x4*)(p0 + (long)(p + 4) * stride);
            const f32x4 c = *(const f32x4*)(p0 + (long)(p + 8) * stride), e = *(const f32x4*)(p0 + (long)(p + 12) * stride);
            s += a;
            s += b;
            s += c;
            s += e;
        }
        for (; p < n_part; p += 4) s += *(const f32x4*)(p0 + (long)p * stride);
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && i4 < n4) {
        f32x4 t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (accumulate) t += *(const f32x4*)(out + i4 * 4);
        *(f32x4*)(out + i4 * 4) = t;
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// One wave per row (statistics recomputed from the saved LN input, two-pass as in the forward kernel):
//   xhat = (x - mean) rstd;  g = dy * gamma;  dx = rstd (g - mean_c(g) - xhat mean_c(g xhat))       [+= if accumulate]
// and per-workgroup partial sums of dgamma = sum_rows dy xhat, dbeta = sum_rows dy (part [gridDim.x][2][C], reduced by
// mage_sum_partials: fixed order, deterministic).
// dxb (optional): the updated dx row once more as bf16 through the dropout mask of the branch it enters next, keep(i) / (1 - p) of
// mage_dropout on the flat index i = row * C + c (thresh 0: a plain cast) -- the operand of that branch's data- and weight-gradient
// GEMMs, which a separate mage_dropout pass (read 4 B + write 2 B per element) produced before.
template <typename DT, int VPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const DT* __restrict__ dy, float* __restrict__ dx, float* __restrict__ part,
                                                            long rows, int C, float eps, int accumulate, unsigned short* __restrict__ dxb,
                                                            unsigned thresh, float inv_keep, unsigned long long seedmul) {
    __shared__ float red[4][2][VPL * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 dg[VPL], db[VPL], gm[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        dg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = j * 256 + lane * 4;
        gm[j] = c < C ? *(const f32x4*)(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float* xr = x + row * C;
        const DT* dr = dy + row * C;
        f32x4 v[VPL], d[VPL];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int c = j * 256 + lane * 4;
            v[j] = c < C ? *(const f32x4*)(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            d[j] = c < C ? load4(dr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int c = j * 256 + lane * 4;
            if (c < C) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = v[j][e] - mean;
                    q += t * t;
                }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int c = j * 256 + lane * 4;
            if (c < C) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (v[j][e] - mean) * rstd;
                    const float g = d[j][e] * gm[j][e];
                    sg += g;
                    sgx += g * xh;
                    dg[j][e] += d[j][e] * xh;
                    db[j][e] += d[j][e];
                    v[j][e] = xh;
                    d[j][e] = g;
                }
            }
        }
        const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
        float* dxr = dx + row * C;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int c = j * 256 + lane * 4;
            if (c < C) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rstd * (d[j][e] - mg - v[j][e] * mgx);
                if (accumulate) o += *(const f32x4*)(dxr + c);
                *(f32x4*)(dxr + c) = o;
                if (dxb) {
                    const unsigned long long i = (unsigned long long)(row * C + c);
                    if (thresh) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = hash32(seedmul + i + e) >= thresh ? o[e] * inv_keep : 0.f;
                    }
                    store4(dxb + i, o);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VPL; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[wave][0][j * 256 + lane * 4 + e] = dg[j][e];
            red[wave][1][j * 256 + lane * 4 + e] = db[j][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        part[((long)blockIdx.x * 2 + 0) * C + c] = (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]);
        part[((long)blockIdx.x * 2 + 1) * C + c] = (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]);
    }
}

// ------------------------------------------------------------------------------------------------ activations
template <int ACT>
__device__ __forceinline__ float actf(float v) {
    if (ACT == MAGE_ACT_RELU) return fmaxf(v, 0.f);
    if (ACT == MAGE_ACT_QUICKGELU) return v / (1.f + expf(-1.702f * v));
    if (ACT == MAGE_ACT_GELU_ERF) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    return v;
}
template <int ACT>
__device__ __forceinline__ float actdf(float v) {
    if (ACT == MAGE_ACT_RELU) return v > 0.f ? 1.f : 0.f;
    if (ACT == MAGE_ACT_QUICKGELU) {
        const float s = 1.f / (1.f + expf(-1.702f * v));
        return s * (1.f + 1.702f * v * (1.f - s));
    }
    if (ACT == MAGE_ACT_GELU_ERF)
        return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * expf(-0.5f * v * v);
    if (ACT == MAGE_ACT_TANH) return 1.f - v * v;       // backward only, and v is the OUTPUT y = tanh(.)
    return 1.f;
}
// BWD = 0: y = act(x);  BWD = 1: y = dy * act'(x)   (x = the saved pre-activation)
template <typename T, int ACT, int BWD>
__global__ __launch_bounds__(256) void act_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ y, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const f32x4 v = load4(x + i);
    f32x4 o;
    if (BWD) {
        const f32x4 g = load4(dy + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = g[e] * actdf<ACT>(v[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = actf<ACT>(v[e]);
    }
    store4(y + i, o);
}

// ------------------------------------------------------------------------------------------------ cross entropy backward
// dlogits[i][k] = (softmax(logits[i])[k] - [k == target[i]]) * scale   (scale = upstream gradient / rows), one wave per row
template <typename OT>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, long rows,
                                                     int K, const float* __restrict__ gout, float inv_rows, OT* __restrict__ dl) {
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
    const float scale = gout[0] * inv_rows, inv = 1.0f / s;
    long tg = target[i];
    if (tg < 0 || tg >= K) tg = -1;                  // reported by the forward kernel (mage_cross_entropy)
    OT* o = dl + i * K;
    for (int k = lane; k < K; k += 64) o[k] = from_f32<OT>((expf(p[k] - mx) * inv - (k == tg ? 1.f : 0.f)) * scale);
}

// ------------------------------------------------------------------------------------------------ embedding backward
// dtable[ids[i]][:] += dout[orow(i)][:]   (orow as in mage_embedding; rows with ids[i] == padding_idx are skipped)
template <typename T>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ dout,
                                                            float* __restrict__ dtable, long n, int C, int n_table, long padding_idx,
                                                            long group, long group_stride, long off) {
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    const long id = ids[i];
    if (id < 0 || id >= n_table || id == padding_idx) return;
    const long orow = (i / group) * group_stride + (i % group) + off;
    const T* src = dout + orow * C;
    float* dst = dtable + id * C;
    for (int c = lane; c < C; c += 64) atomicAdd(dst + c, to_f32<T>(src[c]));
}

// The same scatter for small tables (n_table * 256 B of LDS: the 512-entry visual token table): a workgroup owns a 64-channel slice and
// a chunk of the rows, accumulates into its private LDS copy of the slice with LDS atomics, and flushes the non-zero entries with one
// global atomic each: 262144 x 512 global atomics on 512 hot rows (2.1 ms per call) become 64 x 512 x 512.
template <typename T>
__global__ __launch_bounds__(512) void embedding_bwd_lds_kernel(const int64_t* __restrict__ ids, const T* __restrict__ dout,
                                                                float* __restrict__ dtable, long n, int C, int n_table, long padding_idx,
                                                                long group, long group_stride, long off, long rows_per_chunk,
                                                                float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tab = (float*)smem_raw;                                        // [n_table][64]
    const int c0 = blockIdx.y * 64, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int e = threadIdx.x; e < n_table * 64; e += 512) tab[e] = 0.f;
    __syncthreads();
    const long i0 = (long)blockIdx.x * rows_per_chunk, i1 = min(n, i0 + rows_per_chunk);
    // U rows per wave in flight (one row at a time the kernel ran at one global round trip per row: 1 ms per 250 k rows against ~0.1 ms of bytes)
    constexpr int U = 16;
    for (long ib = i0 + wave; ib < i1; ib += 8 * U) {
        long id[U];
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = ib + 8 * u;
            const long ii = i < i1 ? i : i0;                                // the row's address does not wait for its id
            id[u] = ids[ii];
            const bool ok = i < i1 && id[u] >= 0 && id[u] < n_table && id[u] != padding_idx;
            if (!ok) id[u] = -1;
            // 32-bit quotient (n, group < 2^31: host check): the two 64-bit divisions per row were most of this kernel's time
            const unsigned q = (unsigned)ii / (unsigned)group;
            const long orow = (long)q * group_stride + (long)((unsigned)ii - q * (unsigned)group) + off;
            v[u] = to_f32<T>(dout[orow * C + c0 + lane]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (id[u] >= 0) atomicAdd(&tab[id[u] * 64 + lane], v[u]);
    }
    __syncthreads();
    if (part) {                                                           // this chunk's table slice, plain stores: summed in chunk order by
        float* pp = part + (long)blockIdx.x * n_table * C;                // embedding_bwd_reduce_kernel (no global atomics, deterministic)
        for (int e = threadIdx.x; e < n_table * 64; e += 512) pp[(long)(e >> 6) * C + c0 + (e & 63)] = tab[e];
        return;
    }
    for (int e = threadIdx.x; e < n_table * 64; e += 512) {
        const float v = tab[e];
        if (v != 0.f) atomicAdd(dtable + (long)(e >> 6) * C + c0 + (e & 63), v);
    }
}

// The DETERMINISTIC form of the same scatter (scratch given): the workgroup's geometry is unchanged -- a 64-channel slice, a chunk of the rows,
// an LDS table of the slice -- but the WAVES split the slice's channels instead of the chunk's rows: wave w owns channels 8w .. 8w + 7, a wave
// instruction covers 8 rows x 8 channels (32 bytes of bf16 / 64 of fp32 per row), and the 8 rows are added one after the other in ascending
// order (LDS adds of one wave execute in program order; no other wave touches these channels).  Every (code, channel) sum therefore runs
// over the chunk's rows in ascending row order, whatever the scheduling: with the chunk-order sum of embedding_bwd_reduce_kernel the whole
// gradient is a fixed-order fp32 sum, bit-identical from run to run.  (The row-split kernel above let 8 waves race their LDS atomics on the
// same entries: associativity differences of ~1 ulp per step, enough to move a trained model's near-tie token decisions between runs.)
template <typename T>
__global__ __launch_bounds__(512) void embedding_bwd_det_kernel(const int64_t* __restrict__ ids, const T* __restrict__ dout, long n, int C, int n_table,
                                                                long padding_idx, long group, long group_stride, long off, long rows_per_chunk,
                                                                float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tab = (float*)smem_raw;                                        // [n_table][64]
    const int c0 = blockIdx.y * 64, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int e = threadIdx.x; e < n_table * 64; e += 512) tab[e] = 0.f;
    __syncthreads();
    const long i0 = (long)blockIdx.x * rows_per_chunk, i1 = min(n, i0 + rows_per_chunk);
    const int rs = lane >> 3, ch = wave * 8 + (lane & 7);                  // row slot of the instruction, channel inside the slice
    constexpr int U = 8;                                                   // 8-row groups in flight per wave
    for (long ib = i0; ib < i1; ib += 8 * U) {
        long id[U];
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = ib + 8 * u + rs;
            const long ii = i < i1 ? i : i0;
            id[u] = ids[ii];
            const bool ok = i < i1 && id[u] >= 0 && id[u] < n_table && id[u] != padding_idx;
            if (!ok) id[u] = -1;
            const unsigned q = (unsigned)ii / (unsigned)group;
            const long orow = (long)q * group_stride + (long)((unsigned)ii - q * (unsigned)group) + off;
            v[u] = to_f32<T>(dout[orow * C + c0 + ch]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {                                  // the instruction's 8 rows, ascending: one exec-masked LDS add each
                MAGE_DASSERT(id[u] < n_table && ch < 64);
                if (rs == r && id[u] >= 0) atomicAdd(&tab[id[u] * 64 + ch], v[u]);
                __builtin_amdgcn_wave_barrier();                           // keeps the eight adds eight instructions, in this order
            }
        }
    }
    __syncthreads();
    float* pp = part + (long)blockIdx.x * n_table * C;
    for (int e = threadIdx.x; e < n_table * 64; e += 512) pp[(long)(e >> 6) * C + c0 + (e & 63)] = tab[e];
}

// dtable[e] += sum over chunks (ascending) of part[chunk][e]: the flush of embedding_bwd_lds_kernel without global atomics (the 64 chunks'
// atomics on the same 1 MB table were 0.8 of the kernel's 1.0 ms)
__global__ __launch_bounds__(256) void embedding_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dtable, long n_elem, int n_chunk) {
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= n_elem) return;
    f32x4 a = *(const f32x4*)(dtable + e);
    for (int c = 0; c < n_chunk; ++c) a += *(const f32x4*)(part + (long)c * n_elem + e);
    *(f32x4*)(dtable + e) = a;
}

// ------------------------------------------------------------------------------------------------ grouped row sums
// out[g][c] = sum over rows r with (r / div) % mod == g of w(r) * x[r][c],  w(r) = rs ? rs[r / rs_div] : 1.
// Gradients of broadcast row tables (T / H / W positional embeddings, text positions) and of the speed embedding.
template <typename T>
__global__ __launch_bounds__(256) void group_rowsum_kernel(const T* __restrict__ x, long rows, int C, long div, long mod,
                                                           const float* __restrict__ rs, long rs_div, float* __restrict__ out) {
    const long g = blockIdx.x;
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    const long period = div * mod, nper = (rows + period - 1) / period;
    // blockIdx.z = chunk of the (period, row-in-group) index space: partial sums out[chunk][g][c] for mage_sum_partials
    const long total = nper * div, per = (total + gridDim.z - 1) / gridDim.z;
    const long i0 = (long)blockIdx.z * per, i1 = min(total, i0 + per);
    for (long i = i0; i < i1; ++i) {
        const long q = i / div, j = i - q * div;
        const long r = q * period + g * div + j;
        if (r < rows) s += (rs ? rs[r / rs_div] : 1.f) * to_f32<T>(x[r * C + c]);
    }
    out[((long)blockIdx.z * mod + g) * C + c] = s;
}

// ------------------------------------------------------------------------------------------------ attention backward
// Same addressing as mage_attention (strided row sets, head_dim 32, nk <= 64).  Workgroup = (sequence, group of 4 heads), one
// wave per head.  Per block of 64 queries (lane = query): the probabilities P and dS = P (dP - rowsum(P dP)) are recomputed in
// fp32 and parked in LDS, dq is written by the query's lane; then lane = key accumulates dk_j = scale sum_i dS_ij q_i and
// dv_j = sum_i P_ij dO_i over the block in registers (fixed order: deterministic).  K, V, and the block's Q, dO are staged in
// LDS as fp32.
template <typename T>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const mage_attn_desc d, const T* __restrict__ dout, T* __restrict__ dq,
                                                            T* __restrict__ dk, T* __restrict__ dv, int ld_dq, int ld_dk, int ld_dv, int qb) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = blockIdx.y * 4 + wave;
    const int s = blockIdx.x;
    const int nkp = d.nk + 1;
    // per-wave LDS (qb = queries per block = min(64, nq)): K[nk][32] V[nk][32] Q[qb][33] dO[qb][33] P[qb][nkp] dS[qb][nkp]
    const size_t per_wave = (size_t)(2 * d.nk * 32 + 2 * qb * 33 + 2 * qb * nkp) * 4;
    float* ks = (float*)(smem_raw + wave * per_wave);
    float* vs = ks + d.nk * 32;
    float* qs = vs + d.nk * 32;
    float* gs = qs + qb * 33;
    float* ps = gs + qb * 33;
    float* ds = ps + qb * nkp;
    if (h >= d.n_head) return;                       // whole wave; no workgroup barriers below (waves are independent)
    const unsigned drop_thresh = d.drop_p > 0.f ? (unsigned)((double)d.drop_p * 4294967296.0) : 0u;
    const float inv_keep = 1.0f / (1.0f - d.drop_p);
    const int outer = s / d.inner, in = s - outer * d.inner;
    const long q_base = (long)outer * d.q_outer_stride + in;
    const long kv_base = (long)outer * d.kv_outer_stride + in;
    const T* qp = (const T*)d.q;
    const T* kp = (const T*)d.k;
    const T* vp = (const T*)d.v;
    int klen = d.nk;
    if (d.kv_len) klen = min(klen, d.kv_len[s / d.kv_len_div]);
    for (int e = lane; e < d.nk * 32; e += 64) {
        const int j = e >> 5, c = e & 31;
        const long row = kv_base + (long)j * d.kv_axis_stride;
        ks[e] = to_f32<T>(kp[row * d.ldk + h * 32 + c]);
        vs[e] = to_f32<T>(vp[row * d.ldv + h * 32 + c]);
    }
    float dka[32], dva[32];                          // lane = key j (nk <= 64)
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        dka[c] = 0.f;
        dva[c] = 0.f;
    }
    for (int i0 = 0; i0 < d.nq; i0 += qb) {
        const int nb = min(qb, d.nq - i0);
        for (int e = lane; e < nb * 32; e += 64) {   // stage the block's Q and dO
            const int i = e >> 5, c = e & 31;
            const long row = q_base + (long)(i0 + i) * d.q_axis_stride;
            qs[i * 33 + c] = to_f32<T>(qp[row * d.ldq + h * 32 + c]);
            gs[i * 33 + c] = to_f32<T>(dout[row * d.ldo + h * 32 + c]);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the wave's own LDS writes are visible to all its lanes
        if (lane < nb) {
            const int i = i0 + lane;
            const int jmax = d.causal ? min(klen, i + 1 + (d.nk - d.nq)) : klen;
            float q[32], g[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                q[c] = qs[lane * 33 + c];
                g[c] = gs[lane * 33 + c];
            }
            float mx = -INFINITY;
            for (int j = 0; j < d.nk; ++j) {
                float a = -INFINITY;
                if (j < jmax) {
                    a = 0.f;
#pragma unroll
                    for (int c = 0; c < 32; ++c) a += q[c] * ks[j * 32 + c];
                    a *= d.scale;
                }
                ps[lane * nkp + j] = a;
                mx = fmaxf(mx, a);
            }
            float den = 0.f;
            for (int j = 0; j < d.nk; ++j) {
                const float p = j < jmax ? expf(ps[lane * nkp + j] - mx) : 0.f;
                ps[lane * nkp + j] = p;
                den += p;
            }
            const float inv = 1.0f / den;
            float dsum = 0.f;
            for (int j = 0; j < d.nk; ++j) {
                const float p = ps[lane * nkp + j] * inv;
                float dp = 0.f;
#pragma unroll
                for (int c = 0; c < 32; ++c) dp += g[c] * vs[j * 32 + c];
                if (drop_thresh) {                   // P' = P keep / (1 - p): dP = dP' keep / (1 - p); dv below uses P'
                    const unsigned long long idx = (((unsigned long long)s * d.n_head + h) * d.nq + i) * d.nk + j;
                    dp = hash32(d.drop_seed * 0x9e3779b97f4a7c15ULL + idx) >= drop_thresh ? dp * inv_keep : 0.f;
                }
                ps[lane * nkp + j] = p;
                ds[lane * nkp + j] = dp;
                dsum += p * dp;
            }
            float dqa[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) dqa[c] = 0.f;
            for (int j = 0; j < d.nk; ++j) {
                const float t = ps[lane * nkp + j] * (ds[lane * nkp + j] - dsum);
                ds[lane * nkp + j] = t;
#pragma unroll
                for (int c = 0; c < 32; ++c) dqa[c] += t * ks[j * 32 + c];
            }
            T* o = dq + (q_base + (long)i * d.q_axis_stride) * ld_dq + h * 32;
#pragma unroll
            for (int c = 0; c < 32; ++c) o[c] = from_f32<T>(dqa[c] * d.scale);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (lane < d.nk) {
            for (int i = 0; i < nb; ++i) {
                const float t = ds[i * nkp + lane];
                float p = ps[i * nkp + lane];
                if (drop_thresh) {
                    const unsigned long long idx = (((unsigned long long)s * d.n_head + h) * d.nq + (i0 + i)) * d.nk + lane;
                    p = hash32(d.drop_seed * 0x9e3779b97f4a7c15ULL + idx) >= drop_thresh ? p * inv_keep : 0.f;
                }
#pragma unroll
                for (int c = 0; c < 32; ++c) {
                    dka[c] += t * qs[i * 33 + c];
                    dva[c] += p * gs[i * 33 + c];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < d.nk) {
        const long row = kv_base + (long)lane * d.kv_axis_stride;
        T* ok = dk + row * ld_dk + h * 32;
        T* ov = dv + row * ld_dv + h * 32;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            ok[c] = from_f32<T>(dka[c] * d.scale);
            ov[c] = from_f32<T>(dva[c]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ dropout
// Stateless mask: keep(i) = hash(seed, i) >= p * 2^32, recomputed identically in the backward pass (no mask tensor).
// y = (accumulate ? y : 0) + x * keep / (1 - p)      x: T, y: YT
template <typename T, typename YT>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, YT* __restrict__ y, long n, unsigned thresh, float inv_keep,
                                                      unsigned long long seed, int accumulate) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const f32x4 v = load4(x + i);
    f32x4 o = accumulate ? load4(y + i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (hash32(seed * 0x9e3779b97f4a7c15ULL + (unsigned long long)(i + e)) >= thresh) o[e] += v[e] * inv_keep;
    store4(y + i, o);
}

// ------------------------------------------------------------------------------------------------ Adam
// torch.optim.Adam semantics (main_mage.py:121: betas (0.9, 0.98), eps 1e-6, no weight decay, no amsgrad):
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, float grad_scale) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        f32x4 pp = *(f32x4*)(p + i), gg = *(const f32x4*)(g + i), mm = *(f32x4*)(m + i), vv = *(f32x4*)(v + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gg[e] * grad_scale;
            mm[e] = b1 * mm[e] + (1.f - b1) * ge;
            vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
            pp[e] -= (lr / bc1) * (mm[e] / (sqrtf(vv[e]) / bc2_sqrt + eps));
        }
        *(f32x4*)(p + i) = pp;
        *(f32x4*)(m + i) = mm;
        *(f32x4*)(v + i) = vv;
    } else {
        for (long j = i; j < n; ++j) {
            const float ge = g[j] * grad_scale;
            m[j] = b1 * m[j] + (1.f - b1) * ge;
            v[j] = b2 * v[j] + (1.f - b2) * ge * ge;
            p[j] -= (lr / bc1) * (m[j] / (sqrtf(v[j]) / bc2_sqrt + eps));
        }
    }
}


// ------------------------------------------------------------------------------------------------ BatchNorm (training mode)
// Stage-1 VQ-VAE training (train_vqvae.py:13-35) runs its BatchNorm2d layers on BATCH statistics.  Channels-last rows [rows, C]:
// the statistics are column reductions.  One workgroup owns a slab of rows, thread = channel (coalesced rows), partial sums per
// workgroup [gridDim.x][NOUT][C] are added in a fixed order by mage_sum_partials.
//   MODE 0: sum x                      MODE 1: sum (x - mean)^2                (two-pass variance)
//   MODE 2: sum g, sum g * xhat        g = dy * (mask ? mask > 0 : 1),  xhat = (x - mean) * rstd      (backward)
template <int MODE>
__global__ __launch_bounds__(256) void bn_colreduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ mask, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, long rows, int C, long rows_per_blk,
                                                           float* __restrict__ part) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const long r0 = (long)blockIdx.x * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
    float a0 = 0.f, a1 = 0.f;
    const float m = MODE >= 1 ? mean[c] : 0.f, rs = MODE == 2 ? rstd[c] : 0.f;
    for (long r = r0; r < r1; ++r) {
        const float v = x[r * C + c];
        if (MODE == 0) a0 += v;
        if (MODE == 1) a0 += (v - m) * (v - m);
        if (MODE == 2) {
            float g = dy[r * C + c];
            if (mask && !(mask[r * C + c] > 0.f)) g = 0.f;
            a0 += g;
            a1 += g * ((v - m) * rs);
        }
    }
    constexpr int NOUT = MODE == 2 ? 2 : 1;
    part[((long)blockIdx.x * NOUT + 0) * C + c] = a0;
    if (MODE == 2) part[((long)blockIdx.x * NOUT + 1) * C + c] = a1;
}

// y = [relu]((x - mean) * rstd * gamma + beta [+ residual])
template <typename OT>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ residual, OT* __restrict__ y, long n, int C, int relu) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const int c = (int)(i % C);
    const f32x4 v = *(const f32x4*)(x + i), m = *(const f32x4*)(mean + c), rs = *(const f32x4*)(rstd + c), g = *(const f32x4*)(gamma + c),
                b = *(const f32x4*)(beta + c);
    f32x4 o = (v - m) * rs * g + b;
    if (residual) o += *(const f32x4*)(residual + i);
    if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    store4(y + i, o);
}

// dx = gamma * rstd * (g - s1 / rows - xhat * s2 / rows),  g = dy * (mask ? mask > 0 : 1), (s1, s2) = the MODE 2 column sums
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mask,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ sums, float inv_rows,
                                                           float* __restrict__ dx, long n, int C) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const int c = (int)(i % C);
    const f32x4 v = *(const f32x4*)(x + i), m = *(const f32x4*)(mean + c), rs = *(const f32x4*)(rstd + c), gm = *(const f32x4*)(gamma + c),
                s1 = *(const f32x4*)(sums + c), s2 = *(const f32x4*)(sums + C + c);
    f32x4 g = *(const f32x4*)(dy + i);
    if (mask) {
        const f32x4 mk = *(const f32x4*)(mask + i);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (!(mk[e] > 0.f)) g[e] = 0.f;
    }
    const f32x4 xh = (v - m) * rs;
    *(f32x4*)(dx + i) = gm * rs * (g - s1 * inv_rows - xh * (s2 * inv_rows));
}

// Backward of the decoder head's fold + tanh (mage_convt_fold_tanh): ds = g * (1 - y^2) per output pixel (NCHW), scattered back to
// the per-input-pixel tap products:  dtaps[(n, iy, ix), (ky*4 + kx)*cout + co] = ds[n, co, 2 iy - 1 + ky, 2 ix - 1 + kx]  (0 outside)
__global__ __launch_bounds__(256) void convt_unfold_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ dtaps,
                                                           int N, int IH, int IW, int cout) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int ld = 16 * cout;
    if (gid >= (long)N * IH * IW * ld) return;
    const int col = (int)(gid % ld), co = col % cout, tap = col / cout, ky = tap >> 2, kx = tap & 3;
    const long pix = gid / ld;
    const int ix = (int)(pix % IW), iy = (int)((pix / IW) % IH), n = (int)(pix / ((long)IW * IH));
    const int oy = 2 * iy - 1 + ky, ox = 2 * ix - 1 + kx, OH = 2 * IH, OW = 2 * IW;
    float v = 0.f;
    if ((unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW) {
        const long o = (((long)n * cout + co) * OH + oy) * OW + ox;
        v = y ? g[o] * (1.f - y[o] * y[o]) : g[o];
    }
    dtaps[gid] = v;
}

template <typename T>
int transpose_launch(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t y_row0, int64_t M, int64_t Mp, int32_t C, int32_t out_h,
                     int32_t out_w, int32_t in_h, int32_t in_w, int64_t img_stride, int64_t a_off, int32_t dy, int32_t dx, int32_t stride,
                     hipStream_t s) {
    const dim3 grid((unsigned)((Mp + 63) / 64), (unsigned)((C + 63) / 64));
    if constexpr (sizeof(T) == 2) {
        if (C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && Mp % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
            hipLaunchKernelGGL(transpose8_kernel, grid, dim3(256), 0, s, (const unsigned short*)x, (long)ldx, (unsigned short*)y, (long)ldy,
                               (long)y_row0, (long)M, (long)Mp, C, out_h, out_w, in_h, in_w, (long)img_stride, (long)a_off, dy, dx, stride);
            MAGE_CHECK_LAUNCH("mage_transpose");
            return MAGE_OK;
        }
    }
    hipLaunchKernelGGL((transpose_kernel<T>), grid, dim3(256), 0, s, (const T*)x, (long)ldx, (T*)y, (long)ldy, (long)y_row0, (long)M,
                       (long)Mp, C, out_h, out_w, in_h, in_w, (long)img_stride, (long)a_off, dy, dx, stride);
    MAGE_CHECK_LAUNCH("mage_transpose");
    return MAGE_OK;
}

template <typename DT>
int ln_bwd_launch(const float* x, const float* gamma, const void* dy, float* dx, float* part, int32_t n_part, int64_t rows, int32_t C,
                  float eps, int32_t accumulate, void* dxb, float p, uint64_t seed, hipStream_t s) {
    const dim3 grid(n_part), blk(256);
    const unsigned thresh = (unsigned)((double)p * 4294967296.0);
    const float inv_keep = 1.0f / (1.0f - p);
    const unsigned long long seedmul = (unsigned long long)seed * 0x9e3779b97f4a7c15ULL;
#define LNB(V) hipLaunchKernelGGL((layernorm_bwd_kernel<DT, V>), grid, blk, 0, s, x, gamma, (const DT*)dy, dx, part, (long)rows, C, eps, accumulate, \
                                  (unsigned short*)dxb, thresh, inv_keep, seedmul)
    if (C <= 256) LNB(1);
    else if (C <= 512) LNB(2);
    else if (C <= 1024) LNB(4);
    else LNB(8);
#undef LNB
    MAGE_CHECK_LAUNCH("mage_layernorm_bwd");
    return MAGE_OK;
}

template <typename T, int BWD>
int act_launch(const void* x, const void* dy, void* y, int64_t n, int32_t act, hipStream_t s) {
    const dim3 grid((unsigned)((n / 4 + 255) / 256)), blk(256);
    switch (act) {
        case MAGE_ACT_RELU: hipLaunchKernelGGL((act_kernel<T, MAGE_ACT_RELU, BWD>), grid, blk, 0, s, (const T*)x, (const T*)dy, (T*)y, (long)n); break;
        case MAGE_ACT_QUICKGELU: hipLaunchKernelGGL((act_kernel<T, MAGE_ACT_QUICKGELU, BWD>), grid, blk, 0, s, (const T*)x, (const T*)dy, (T*)y, (long)n); break;
        case MAGE_ACT_GELU_ERF: hipLaunchKernelGGL((act_kernel<T, MAGE_ACT_GELU_ERF, BWD>), grid, blk, 0, s, (const T*)x, (const T*)dy, (T*)y, (long)n); break;
        case MAGE_ACT_TANH:
            if (BWD) { hipLaunchKernelGGL((act_kernel<T, MAGE_ACT_TANH, BWD>), grid, blk, 0, s, (const T*)x, (const T*)dy, (T*)y, (long)n); break; }
            [[fallthrough]];
        default: mage_set_error("mage_act: activation %d unsupported", act); return MAGE_EINVAL;
    }
    MAGE_CHECK_LAUNCH("mage_act");
    return MAGE_OK;
}

}  // namespace

#define DT_DISPATCH(dtype, CALL_F32, CALL_BF16, who)                  \
    do {                                                              \
        if ((dtype) == MAGE_F32) return CALL_F32;                     \
        if ((dtype) == MAGE_BF16) return CALL_BF16;                   \
        mage_set_error("%s: bad dtype %d", who, (int)(dtype));        \
        return MAGE_EINVAL;                                           \
    } while (0)

extern "C" int mage_transpose(const void* x, int32_t dtype, int64_t ldx, void* y, int64_t ldy, int64_t y_row0, int64_t M, int64_t Mp,
                              int32_t C, int32_t out_h, int32_t out_w, int32_t in_h, int32_t in_w, int64_t img_stride, int64_t a_off,
                              int32_t dy, int32_t dx, int32_t stride, void* stream) {
    MAGE_CHECK_ARG(x && y && M > 0 && Mp >= M && C > 0 && out_h >= 1 && out_w >= 1 && in_h >= 1 && in_w >= 1 && ldy >= Mp && stride >= 1,
                   "mage_transpose: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DT_DISPATCH(dtype, (transpose_launch<float>(x, ldx, y, ldy, y_row0, M, Mp, C, out_h, out_w, in_h, in_w, img_stride, a_off, dy, dx, stride, s)),
                (transpose_launch<unsigned short>(x, ldx, y, ldy, y_row0, M, Mp, C, out_h, out_w, in_h, in_w, img_stride, a_off, dy, dx, stride, s)),
                "mage_transpose");
}


extern "C" int mage_transpose_colsum(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t M, int64_t Mp, int32_t C, int32_t out_w,
                                     int64_t img_stride, int64_t a_off, float* colsum, int32_t n_part, void* stream) {
    const long tiles = (Mp + 63) / 64;
    MAGE_CHECK_ARG(x && y && colsum && M > 0 && Mp >= M && C > 0 && out_w >= 1 && ldy >= Mp, "mage_transpose_colsum: bad arguments");
    MAGE_CHECK_ARG(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && Mp % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0,
                   "mage_transpose_colsum: bf16 operands with 16-byte rows");
    MAGE_CHECK_ARG(n_part == (int)((tiles + TRANSPOSE_TPW - 1) / TRANSPOSE_TPW), "mage_transpose_colsum: n_part must be ceil(ceil(Mp/64)/%d)",
                   TRANSPOSE_TPW);
    // rows regrouped like mage_transpose with out_h = 1 (row m -> (m / out_w) * img_stride + m % out_w + a_off), no tap shift
    hipLaunchKernelGGL(transpose8_colsum_kernel, dim3(n_part, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x,
                       (long)ldx, (unsigned short*)y, (long)ldy, (long)M, (long)Mp, C, 1, out_w, 1, out_w, (long)img_stride, (long)a_off, colsum);
    MAGE_CHECK_LAUNCH("mage_transpose_colsum");
    return MAGE_OK;
}

extern "C" int mage_bn_colreduce(int32_t mode, const float* x, const float* dy, const float* mask, const float* mean, const float* rstd,
                                 int64_t rows, int32_t C, float* partials, int32_t n_part, void* stream) {
    MAGE_CHECK_ARG(x && partials && rows > 0 && C > 0 && n_part >= 1 && mode >= 0 && mode <= 2, "mage_bn_colreduce: bad arguments");
    MAGE_CHECK_ARG(mode == 0 || mean, "mage_bn_colreduce: mean missing");
    MAGE_CHECK_ARG(mode != 2 || (dy && rstd), "mage_bn_colreduce: dy / rstd missing");
    const long rpb = (rows + n_part - 1) / n_part;
    const dim3 grid(n_part, (C + 255) / 256), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL((bn_colreduce_kernel<0>), grid, blk, 0, s, x, dy, mask, mean, rstd, (long)rows, C, rpb, partials);
    else if (mode == 1) hipLaunchKernelGGL((bn_colreduce_kernel<1>), grid, blk, 0, s, x, dy, mask, mean, rstd, (long)rows, C, rpb, partials);
    else hipLaunchKernelGGL((bn_colreduce_kernel<2>), grid, blk, 0, s, x, dy, mask, mean, rstd, (long)rows, C, rpb, partials);
    MAGE_CHECK_LAUNCH("mage_bn_colreduce");
    return MAGE_OK;
}

extern "C" int mage_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                             const float* residual, void* y, int32_t y_dtype, int64_t rows, int32_t C, int32_t relu, void* stream) {
    MAGE_CHECK_ARG(x && mean && rstd && gamma && beta && y && rows > 0 && C > 0 && C % 4 == 0, "mage_bn_apply: bad arguments");
    const long n = (long)rows * C;
    const dim3 grid((unsigned)((n / 4 + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (y_dtype == MAGE_F32) hipLaunchKernelGGL((bn_apply_kernel<float>), grid, blk, 0, s, x, mean, rstd, gamma, beta, residual, (float*)y, n, C, relu);
    else if (y_dtype == MAGE_BF16)
        hipLaunchKernelGGL((bn_apply_kernel<unsigned short>), grid, blk, 0, s, x, mean, rstd, gamma, beta, residual, (unsigned short*)y, n, C, relu);
    else { mage_set_error("mage_bn_apply: bad y_dtype %d", y_dtype); return MAGE_EINVAL; }
    MAGE_CHECK_LAUNCH("mage_bn_apply");
    return MAGE_OK;
}

extern "C" int mage_bn_bwd_apply(const float* x, const float* dy, const float* mask, const float* mean, const float* rstd, const float* gamma,
                                 const float* sums, float* dx, int64_t rows, int32_t C, void* stream) {
    MAGE_CHECK_ARG(x && dy && mean && rstd && gamma && sums && dx && rows > 0 && C > 0 && C % 4 == 0, "mage_bn_bwd_apply: bad arguments");
    const long n = (long)rows * C;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, mask, mean, rstd,
                       gamma, sums, 1.0f / (float)rows, dx, n, C);
    MAGE_CHECK_LAUNCH("mage_bn_bwd_apply");
    return MAGE_OK;
}

extern "C" int mage_convt_unfold_tanh_bwd(const float* grad_y, const float* y, float* dtaps, int32_t N, int32_t IH, int32_t IW, int32_t cout,
                                          void* stream) {
    MAGE_CHECK_ARG(grad_y && dtaps && N > 0 && IH > 0 && IW > 0 && cout >= 1 && cout <= 4, "mage_convt_unfold_tanh_bwd: bad arguments");
    const long total = (long)N * IH * IW * 16 * cout;
    hipLaunchKernelGGL(convt_unfold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_y, y, dtaps, N, IH,
                       IW, cout);
    MAGE_CHECK_LAUNCH("mage_convt_unfold_tanh_bwd");
    return MAGE_OK;
}

extern "C" int mage_row_sum(const void* x, int32_t dtype, int64_t ld, int64_t n, int32_t rows, float* out, int32_t n_chunk, void* stream) {
    MAGE_CHECK_ARG(x && out && rows > 0 && n > 0 && n_chunk >= 1, "mage_row_sum: bad arguments");
    const dim3 grid((rows + 3) / 4, n_chunk), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MAGE_F32) hipLaunchKernelGGL((row_sum_kernel<float>), grid, blk, 0, s, (const float*)x, (long)ld, (long)n, rows, out);
    else if (dtype == MAGE_BF16) hipLaunchKernelGGL((row_sum_kernel<unsigned short>), grid, blk, 0, s, (const unsigned short*)x, (long)ld, (long)n, rows, out);
    else { mage_set_error("mage_row_sum: bad dtype %d", dtype); return MAGE_EINVAL; }
    MAGE_CHECK_LAUNCH("mage_row_sum");
    return MAGE_OK;
}

extern "C" int mage_sum_partials(const float* part, int64_t stride, int32_t n_part, int64_t n, float* out, int32_t accumulate, void* stream) {
    MAGE_CHECK_ARG(part && out && n_part > 0 && n > 0, "mage_sum_partials: bad arguments");
    if (n % 4 == 0 && stride % 4 == 0 && n_part >= 4 && ((((uintptr_t)part | (uintptr_t)out) & 15) == 0)) {
        hipLaunchKernelGGL(sum_partials4_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, (hipStream_t)stream, part, (long)stride, n_part,
                           (long)(n / 4), out, accumulate);
        MAGE_CHECK_LAUNCH("mage_sum_partials");
        return MAGE_OK;
    }
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, (long)stride, n_part,
                       (long)n, out, accumulate);
    MAGE_CHECK_LAUNCH("mage_sum_partials");
    return MAGE_OK;
}

extern "C" int mage_layernorm_bwd(const float* x, const float* gamma, const void* dy, int32_t dy_dtype, float* dx, float* partials,
                                  int32_t n_part, int64_t rows, int32_t C, float eps, int32_t accumulate, void* dx_bf16, float p, uint64_t seed,
                                  void* stream) {
    MAGE_CHECK_ARG(x && gamma && dy && dx && partials, "mage_layernorm_bwd: null pointer");
    MAGE_CHECK_ARG(p >= 0.f && p < 1.f && (!dx_bf16 || ((uintptr_t)dx_bf16 & 7) == 0), "mage_layernorm_bwd: bad dx_bf16 / p");
    MAGE_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0 && C <= 2048 && n_part >= 1, "mage_layernorm_bwd: rows=%ld C=%d unsupported", (long)rows, C);
    hipStream_t s = (hipStream_t)stream;
    DT_DISPATCH(dy_dtype, (ln_bwd_launch<float>(x, gamma, dy, dx, partials, n_part, rows, C, eps, accumulate, dx_bf16, p, seed, s)),
                (ln_bwd_launch<unsigned short>(x, gamma, dy, dx, partials, n_part, rows, C, eps, accumulate, dx_bf16, p, seed, s)),
                "mage_layernorm_bwd");
}

extern "C" int mage_act(const void* x, void* y, int32_t dtype, int64_t n, int32_t act, void* stream) {
    MAGE_CHECK_ARG(x && y && n > 0 && n % 4 == 0, "mage_act: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DT_DISPATCH(dtype, (act_launch<float, 0>(x, nullptr, y, n, act, s)), (act_launch<unsigned short, 0>(x, nullptr, y, n, act, s)), "mage_act");
}

extern "C" int mage_act_bwd(const void* x, const void* dy, void* dx, int32_t dtype, int64_t n, int32_t act, void* stream) {
    MAGE_CHECK_ARG(x && dy && dx && n > 0 && n % 4 == 0, "mage_act_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DT_DISPATCH(dtype, (act_launch<float, 1>(x, dy, dx, n, act, s)), (act_launch<unsigned short, 1>(x, dy, dx, n, act, s)), "mage_act_bwd");
}

extern "C" int mage_cross_entropy_bwd(const float* logits, const int64_t* target, int64_t rows, int32_t K, const float* grad_out,
                                      void* dlogits, int32_t dl_dtype, void* stream) {
    MAGE_CHECK_ARG(logits && target && grad_out && dlogits && rows > 0 && K > 0, "mage_cross_entropy_bwd: bad arguments");
    const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (dl_dtype == MAGE_F32)
        hipLaunchKernelGGL((ce_bwd_kernel<float>), grid, blk, 0, s, logits, target, (long)rows, K, grad_out, 1.0f / (float)rows, (float*)dlogits);
    else if (dl_dtype == MAGE_BF16)
        hipLaunchKernelGGL((ce_bwd_kernel<unsigned short>), grid, blk, 0, s, logits, target, (long)rows, K, grad_out, 1.0f / (float)rows,
                           (unsigned short*)dlogits);
    else { mage_set_error("mage_cross_entropy_bwd: bad dtype %d", dl_dtype); return MAGE_EINVAL; }
    MAGE_CHECK_LAUNCH("mage_cross_entropy_bwd");
    return MAGE_OK;
}

extern "C" int mage_embedding_bwd(const int64_t* ids, const void* dout, int32_t dout_dtype, float* dtable, int64_t n, int32_t C,
                                  int32_t n_table, int64_t padding_idx, int64_t group, int64_t group_stride, int64_t off, float* scratch,
                                  int64_t scratch_floats, void* stream) {
    MAGE_CHECK_ARG(ids && dout && dtable && n > 0 && C > 0 && n_table > 0 && group > 0, "mage_embedding_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const bool small_tab = n_table <= 512 && C % 64 == 0 && n < (1L << 31) && group < (1L << 31) && (dout_dtype == MAGE_F32 || dout_dtype == MAGE_BF16);
    if (small_tab && (n >= 8192 || scratch)) {
        const int n_chunk = (int)(n / 4096 < 64 ? (n + 4095) / 4096 : 64);
        const long rpc = (((n + n_chunk - 1) / n_chunk) + 7) & ~7L;       // whole 8-row groups per chunk (the deterministic kernel's unit)
        const dim3 grid(n_chunk, C / 64), blk(512);
        const size_t lds = (size_t)n_table * 64 * sizeof(float);
        float* part = (scratch && scratch_floats >= (int64_t)n_chunk * n_table * C && (((uintptr_t)scratch | (uintptr_t)dtable) & 15) == 0) ? scratch : nullptr;
        if (part) {                                                       // deterministic: channel-split waves, rows in ascending order, chunk-order sum
            static bool attr_det[MAGE_MAX_DEVICES][2] = {{false}};
            const int dev_ = mage_device_index();
            MAGE_CHECK_ARG(dev_ >= 0, "mage_embedding_bwd: no current device");
            const int ti = dout_dtype == MAGE_F32 ? 0 : 1;
            if (!attr_det[dev_][ti]) {
                if (ti == 0) (void)hipFuncSetAttribute((const void*)embedding_bwd_det_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                else (void)hipFuncSetAttribute((const void*)embedding_bwd_det_kernel<unsigned short>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                attr_det[dev_][ti] = true;
            }
            if (ti == 0)
                hipLaunchKernelGGL((embedding_bwd_det_kernel<float>), grid, blk, lds, s, ids, (const float*)dout, (long)n, C, n_table, (long)padding_idx,
                                   (long)group, (long)group_stride, (long)off, rpc, part);
            else
                hipLaunchKernelGGL((embedding_bwd_det_kernel<unsigned short>), grid, blk, lds, s, ids, (const unsigned short*)dout, (long)n, C, n_table,
                                   (long)padding_idx, (long)group, (long)group_stride, (long)off, rpc, part);
            const long n_elem = (long)n_table * C;
            hipLaunchKernelGGL(embedding_bwd_reduce_kernel, dim3((unsigned)((n_elem / 4 + 255) / 256)), dim3(256), 0, s, part, dtable, n_elem, n_chunk);
            MAGE_CHECK_LAUNCH("mage_embedding_bwd");
            return MAGE_OK;
        }
        static bool attr_set[MAGE_MAX_DEVICES][2] = {{false}};
        const int dev = mage_device_index();
        MAGE_CHECK_ARG(dev >= 0, "mage_embedding_bwd: no current device");
        if (dout_dtype == MAGE_F32) {
            if (!attr_set[dev][0]) {
                (void)hipFuncSetAttribute((const void*)embedding_bwd_lds_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                attr_set[dev][0] = true;
            }
            hipLaunchKernelGGL((embedding_bwd_lds_kernel<float>), grid, blk, lds, s, ids, (const float*)dout, dtable, (long)n, C, n_table,
                               (long)padding_idx, (long)group, (long)group_stride, (long)off, rpc, part);
        } else {
            if (!attr_set[dev][1]) {
                (void)hipFuncSetAttribute((const void*)embedding_bwd_lds_kernel<unsigned short>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          128 * 1024);
                attr_set[dev][1] = true;
            }
            hipLaunchKernelGGL((embedding_bwd_lds_kernel<unsigned short>), grid, blk, lds, s, ids, (const unsigned short*)dout, dtable, (long)n, C,
                               n_table, (long)padding_idx, (long)group, (long)group_stride, (long)off, rpc, part);
        }
        if (part) {
            const long n_elem = (long)n_table * C;
            hipLaunchKernelGGL(embedding_bwd_reduce_kernel, dim3((unsigned)((n_elem / 4 + 255) / 256)), dim3(256), 0, s, part, dtable, n_elem, n_chunk);
        }
        MAGE_CHECK_LAUNCH("mage_embedding_bwd");
        return MAGE_OK;
    }
    const dim3 grid((unsigned)((n + 3) / 4)), blk(256);
    if (dout_dtype == MAGE_F32)
        hipLaunchKernelGGL((embedding_bwd_kernel<float>), grid, blk, 0, s, ids, (const float*)dout, dtable, (long)n, C, n_table, (long)padding_idx,
                           (long)group, (long)group_stride, (long)off);
    else if (dout_dtype == MAGE_BF16)
        hipLaunchKernelGGL((embedding_bwd_kernel<unsigned short>), grid, blk, 0, s, ids, (const unsigned short*)dout, dtable, (long)n, C, n_table,
                           (long)padding_idx, (long)group, (long)group_stride, (long)off);
    else { mage_set_error("mage_embedding_bwd: bad dtype %d", dout_dtype); return MAGE_EINVAL; }
    MAGE_CHECK_LAUNCH("mage_embedding_bwd");
    return MAGE_OK;
}

extern "C" int mage_group_rowsum(const void* x, int32_t dtype, int64_t rows, int32_t C, int64_t div, int64_t mod, const float* row_scale,
                                 int64_t row_scale_div, float* out, int32_t n_chunk, void* stream) {
    MAGE_CHECK_ARG(x && out && rows > 0 && C > 0 && div >= 1 && mod >= 1 && n_chunk >= 1 && n_chunk <= 65535 && (!row_scale || row_scale_div >= 1),
                   "mage_group_rowsum: bad arguments");
    const dim3 grid((unsigned)mod, (C + 255) / 256, n_chunk), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MAGE_F32)
        hipLaunchKernelGGL((group_rowsum_kernel<float>), grid, blk, 0, s, (const float*)x, (long)rows, C, (long)div, (long)mod, row_scale,
                           (long)row_scale_div, out);
    else if (dtype == MAGE_BF16)
        hipLaunchKernelGGL((group_rowsum_kernel<unsigned short>), grid, blk, 0, s, (const unsigned short*)x, (long)rows, C, (long)div, (long)mod,
                           row_scale, (long)row_scale_div, out);
    else { mage_set_error("mage_group_rowsum: bad dtype %d", dtype); return MAGE_EINVAL; }
    MAGE_CHECK_LAUNCH("mage_group_rowsum");
    return MAGE_OK;
}

namespace {
// ------------------------------------------------------------------------------------------------ attention backward on the matrix cores
// bf16, nq <= 32, nk <= 32 (the decoder's axial attentions): the backward twin of attention_mfma_kernel (norm_attn.hip), one wave per
// (sequence, head), 4 heads per workgroup.  K, Q, dO of the head are staged in wave-private LDS (bf16, 80-byte rows); V fragments come
// straight from global.  Per 16-query block and 16-key block, with the MFMA conventions of the forward kernel
// (16x16x32: D[m][n] = sum_c A[m][c] B[n][c], lane (r, g) feeds row r's bytes [16g, 16g+16) of A and of B, result lane (n, g) holds
//  D[4g+e][n];  16x16x16: A lane (m, g) holds A[m][4g+e], B lane (n, g) holds B[n][4g+e], result lane (n, g) holds D[4g+e][n]):
//   query-major  S^T = K Q^T, dP^T = V dO^T: lane (i, g) holds 4 keys of ITS query: softmax statistics (max, sum) and D_i = sum_j P dP
//                as in-lane sums + two xor-shuffles; dS = P (dP - D) scale;  dQ^T = K^T dS^T  (A = K^T read from LDS, B = dS in the lane);
//   key-major    S = Q K^T, dP = dO V^T: lane (j, g) holds 4 queries of ITS key; P and dS are rebuilt with the statistics fetched from
//                the lanes that own those queries;  dV^T += dO^T P, dK^T += Q^T dS  (A = dO^T / Q^T from LDS).
// P and dS enter the 16x16x16 MFMAs as bf16 hi + lo pairs (fp32-class weights, as in the forward).  dK / dV accumulate over the query
// blocks in registers; every output row leaves as one 16-byte store per lane (lane-group swap as in the forward).
typedef __attribute__((ext_vector_type(8))) __bf16 tbf16x8;
typedef __attribute__((ext_vector_type(4))) short tshort4;

__device__ __forceinline__ void split_hi_lo(float v, short& hi, short& lo) {
    const unsigned hb = __float_as_uint(v) & 0xffff0000u;
    hi = (short)(hb >> 16);
    lo = (short)(__float_as_uint(v - __uint_as_float(hb)) >> 16);
}

template <int NKB>
__global__ __launch_bounds__(256) void attention_bwd_mfma_kernel(const mage_attn_desc d, const unsigned short* __restrict__ dout,
                                                                 unsigned short* __restrict__ dq, unsigned short* __restrict__ dk,
                                                                 unsigned short* __restrict__ dv, int ld_dq, int ld_dk, int ld_dv) {
    constexpr int PITCH = 40;                                  // shorts per staged row (64 data bytes + 16): conflict-free transposed reads
    __shared__ __attribute__((aligned(16))) unsigned short lds[4][(16 * NKB + 64) * PITCH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = blockIdx.y * 4 + wave;
    if (h >= d.n_head) return;                                 // whole wave; waves are independent (no workgroup barrier below)
    unsigned short* ks = lds[wave];
    unsigned short* qs = ks + 16 * NKB * PITCH;
    unsigned short* gs = qs + 32 * PITCH;
    const int s = blockIdx.x;
    const int outer = s / d.inner, in = s - outer * d.inner;
    const long q_base = (long)outer * d.q_outer_stride + in;
    const long kv_base = (long)outer * d.kv_outer_stride + in;
    const unsigned short* qp = (const unsigned short*)d.q;
    const unsigned short* kp = (const unsigned short*)d.k;
    const unsigned short* vp = (const unsigned short*)d.v;
    const int r = lane & 15, g = lane >> 4;
    const int sr = lane >> 2, sc = (lane & 3) * 8;             // staging: 16 rows x 4 chunks of 16 bytes per pass
    const int nqb = (d.nq + 15) >> 4;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int j = kb * 16 + sr;
        uint4 val = uint4{0u, 0u, 0u, 0u};
        if (j < d.nk) val = *(const uint4*)(kp + (kv_base + (long)j * d.kv_axis_stride) * d.ldk + h * 32 + sc);
        *(uint4*)(ks + j * PITCH + sc) = val;
    }
    for (int qb = 0; qb < nqb; ++qb) {
        const int i = qb * 16 + sr;
        uint4 vq = uint4{0u, 0u, 0u, 0u}, vg = vq;
        if (i < d.nq) {
            const long row = q_base + (long)i * d.q_axis_stride;
            vq = *(const uint4*)(qp + row * d.ldq + h * 32 + sc);
            vg = *(const uint4*)(dout + row * d.ldo + h * 32 + sc);
        }
        *(uint4*)(qs + i * PITCH + sc) = vq;
        *(uint4*)(gs + i * PITCH + sc) = vg;
    }
    uint4 vf[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const long vrow = kv_base + (long)min(kb * 16 + r, d.nk - 1) * d.kv_axis_stride;      // rows >= nk: clamped, masked below
        vf[kb] = *(const uint4*)(vp + vrow * d.ldv + h * 32 + g * 8);
    }
    int klen = d.nk;
    if (d.kv_len) klen = min(klen, d.kv_len[s / d.kv_len_div]);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);                        // lgkmcnt(0): the wave's own LDS writes are visible to all its lanes
    uint4 kf[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) kf[kb] = *(const uint4*)(ks + (kb * 16 + r) * PITCH + g * 8);
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 dka[NKB][2], dva[NKB][2];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int b = 0; b < 2; ++b) dka[kb][b] = dva[kb][b] = zero4;

    for (int qb = 0; qb < nqb; ++qb) {
        const uint4 qf = *(const uint4*)(qs + (qb * 16 + r) * PITCH + g * 8);
        const uint4 gf = *(const uint4*)(gs + (qb * 16 + r) * PITCH + g * 8);
        // ---------------- query-major: this lane's query qi, keys kb*16 + 4g + e
        const int qi = qb * 16 + r;
        const int jmax = d.causal ? min(klen, qi + 1 + (d.nk - d.nq)) : klen;
        f32x4 st[NKB], dpt[NKB];
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tbf16x8, kf[kb]), __builtin_bit_cast(tbf16x8, qf), zero4, 0, 0, 0);
            dpt[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tbf16x8, vf[kb]), __builtin_bit_cast(tbf16x8, gf), zero4, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                st[kb][e] = (kb * 16 + 4 * g + e < jmax) ? st[kb][e] * d.scale : -INFINITY;
                mx = fmaxf(mx, st[kb][e]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float den = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                st[kb][e] = (kb * 16 + 4 * g + e < jmax) ? expf(st[kb][e] - mx) : 0.f;
                den += st[kb][e];
            }
        den += __shfl_xor(den, 16);
        den += __shfl_xor(den, 32);
        const float inv = 1.0f / den;
        float dsum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                st[kb][e] *= inv;                                             // P
                dsum += st[kb][e] * dpt[kb][e];
            }
        dsum += __shfl_xor(dsum, 16);
        dsum += __shfl_xor(dsum, 32);
        tshort4 shi[NKB], slo[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                short hi, lo;
                split_hi_lo(st[kb][e] * (dpt[kb][e] - dsum) * d.scale, hi, lo);    // dS
                shi[kb][e] = hi;
                slo[kb][e] = lo;
            }
        f32x4 o[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            o[b] = zero4;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const unsigned short* kr = ks + (kb * 16 + 4 * g) * PITCH + b * 16 + r;
                tshort4 kt;
#pragma unroll
                for (int e = 0; e < 4; ++e) kt[e] = (short)kr[e * PITCH];
                o[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kt, shi[kb], o[b], 0, 0, 0);
                o[b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kt, slo[kb], o[b], 0, 0, 0);
            }
        }
        {
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(o[0][e]), __float_as_uint(o[1][e]), false, false);
                v0[e] = __uint_as_float(sw[0]);
                v1[e] = __uint_as_float(sw[1]);
            }
            if (qi < d.nq) store8(dq + (q_base + (long)qi * d.q_axis_stride) * ld_dq + h * 32 + 16 * (g & 1) + 8 * (g >> 1), v0, v1);
        }
        // ---------------- key-major: this lane's key kb*16 + r, queries qb*16 + 4g + e (their statistics live in lanes 4g + e)
        float mxq[4], invq[4], dsq[4];
        int jmq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mxq[e] = __shfl(mx, 4 * g + e);
            invq[e] = __shfl(inv, 4 * g + e);
            dsq[e] = __shfl(dsum, 4 * g + e);
            const int i2 = qb * 16 + 4 * g + e;
            jmq[e] = d.causal ? min(klen, i2 + 1 + (d.nk - d.nq)) : klen;
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const f32x4 s2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tbf16x8, qf), __builtin_bit_cast(tbf16x8, kf[kb]), zero4, 0, 0, 0);
            const f32x4 dp2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tbf16x8, gf), __builtin_bit_cast(tbf16x8, vf[kb]), zero4, 0, 0, 0);
            tshort4 phi, plo, dhi, dlo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = (kb * 16 + r < jmq[e]) ? expf(s2[e] * d.scale - mxq[e]) * invq[e] : 0.f;
                short hi, lo;
                split_hi_lo(p, hi, lo);
                phi[e] = hi;
                plo[e] = lo;
                split_hi_lo(p * (dp2[e] - dsq[e]) * d.scale, hi, lo);
                dhi[e] = hi;
                dlo[e] = lo;
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const unsigned short* gr = gs + (qb * 16 + 4 * g) * PITCH + b * 16 + r;
                const unsigned short* qr = qs + (qb * 16 + 4 * g) * PITCH + b * 16 + r;
                tshort4 gt, qt;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gt[e] = (short)gr[e * PITCH];
                    qt[e] = (short)qr[e * PITCH];
                }
                dva[kb][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(gt, phi, dva[kb][b], 0, 0, 0);
                dva[kb][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(gt, plo, dva[kb][b], 0, 0, 0);
                dka[kb][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qt, dhi, dka[kb][b], 0, 0, 0);
                dka[kb][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qt, dlo, dka[kb][b], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        f32x4 k0, k1, w0, w1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto sk = __builtin_amdgcn_permlane16_swap(__float_as_uint(dka[kb][0][e]), __float_as_uint(dka[kb][1][e]), false, false);
            const auto sv = __builtin_amdgcn_permlane16_swap(__float_as_uint(dva[kb][0][e]), __float_as_uint(dva[kb][1][e]), false, false);
            k0[e] = __uint_as_float(sk[0]);
            k1[e] = __uint_as_float(sk[1]);
            w0[e] = __uint_as_float(sv[0]);
            w1[e] = __uint_as_float(sv[1]);
        }
        const int j = kb * 16 + r;
        if (j < d.nk) {
            const long row = kv_base + (long)j * d.kv_axis_stride;
            const int col = h * 32 + 16 * (g & 1) + 8 * (g >> 1);
            store8(dk + row * ld_dk + col, k0, k1);
            store8(dv + row * ld_dv + col, w0, w1);
        }
    }
}

}  // namespace

extern "C" int mage_attention_bwd(const mage_attn_desc* d, const void* dout, void* dq, void* dk, void* dv, int32_t ld_dq, int32_t ld_dk,
                                  int32_t ld_dv, void* stream) {
    MAGE_CHECK_ARG(d && d->q && d->k && d->v && dout && dq && dk && dv, "mage_attention_bwd: null pointer");
    MAGE_CHECK_ARG(d->nk >= 1 && d->nk <= 64 && d->nq >= 1 && d->n_seq >= 1 && d->n_head >= 1 && d->inner >= 1,
                   "mage_attention_bwd: nk=%d nq=%d unsupported", d->nk, d->nq);
    hipStream_t s = (hipStream_t)stream;
    MAGE_CHECK_ARG(d->drop_p >= 0.f && d->drop_p < 1.f && (d->drop_p == 0.f || d->dtype == MAGE_F32),
                   "mage_attention_bwd: drop_p=%g needs the fp32 kernels and 0 <= p < 1", (double)d->drop_p);
    MAGE_CHECK_ARG(d->o_axis_stride == 0 && d->o_outer_stride == 0, "mage_attention_bwd: dout is addressed like q (no separate output row map)");
    if (d->dtype == MAGE_BF16 && d->nq <= 32 && d->nk <= 32 && !mage_options().attn_no_mfma && d->ldq % 8 == 0 && d->ldk % 8 == 0 &&
        d->ldv % 8 == 0 && d->ldo % 8 == 0 && ld_dq % 8 == 0 && ld_dk % 8 == 0 && ld_dv % 8 == 0 &&
        ((((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0)) {
        const dim3 grid(d->n_seq, (d->n_head + 3) / 4), blk(256);
        if (d->nk <= 16)
            hipLaunchKernelGGL((attention_bwd_mfma_kernel<1>), grid, blk, 0, s, *d, (const unsigned short*)dout, (unsigned short*)dq,
                               (unsigned short*)dk, (unsigned short*)dv, ld_dq, ld_dk, ld_dv);
        else
            hipLaunchKernelGGL((attention_bwd_mfma_kernel<2>), grid, blk, 0, s, *d, (const unsigned short*)dout, (unsigned short*)dq,
                               (unsigned short*)dk, (unsigned short*)dv, ld_dq, ld_dk, ld_dv);
        MAGE_CHECK_LAUNCH("mage_attention_bwd");
        return MAGE_OK;
    }
    const int qb = d->nq < 64 ? d->nq : 64;
    const size_t per_wave = (size_t)(2 * d->nk * 32 + 2 * qb * 33 + 2 * qb * (d->nk + 1)) * 4;
    const size_t lds = 4 * per_wave;
    MAGE_CHECK_ARG(lds <= 160 * 1024, "mage_attention_bwd: LDS budget");
    const dim3 grid(d->n_seq, (d->n_head + 3) / 4), blk(256);
    static bool attr_set[MAGE_MAX_DEVICES][2] = {{false}};
    const int dev = mage_device_index();
    MAGE_CHECK_ARG(dev >= 0, "mage_attention_bwd: no current device");
    if (d->dtype == MAGE_F32) {
        if (!attr_set[dev][0]) {
            (void)hipFuncSetAttribute((const void*)attention_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set[dev][0] = true;
        }
        hipLaunchKernelGGL((attention_bwd_kernel<float>), grid, blk, lds, s, *d, (const float*)dout, (float*)dq, (float*)dk, (float*)dv, ld_dq,
                           ld_dk, ld_dv, qb);
    } else if (d->dtype == MAGE_BF16) {
        if (!attr_set[dev][1]) {
            (void)hipFuncSetAttribute((const void*)attention_bwd_kernel<unsigned short>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set[dev][1] = true;
        }
        hipLaunchKernelGGL((attention_bwd_kernel<unsigned short>), grid, blk, lds, s, *d, (const unsigned short*)dout, (unsigned short*)dq,
                           (unsigned short*)dk, (unsigned short*)dv, ld_dq, ld_dk, ld_dv, qb);
    } else {
        mage_set_error("mage_attention_bwd: bad dtype %d", d->dtype);
        return MAGE_EINVAL;
    }
    MAGE_CHECK_LAUNCH("mage_attention_bwd");
    return MAGE_OK;
}

extern "C" int mage_dropout(const void* x, int32_t x_dtype, void* y, int32_t y_dtype, int64_t n, float p, uint64_t seed, int32_t accumulate,
                            void* stream) {
    MAGE_CHECK_ARG(x && y && n > 0 && n % 4 == 0 && p >= 0.f && p < 1.f, "mage_dropout: bad arguments");
    const unsigned thresh = (unsigned)((double)p * 4294967296.0);
    const float inv_keep = 1.0f / (1.0f - p);
    const dim3 grid((unsigned)((n / 4 + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
#define DROP(T, YT) hipLaunchKernelGGL((dropout_kernel<T, YT>), grid, blk, 0, s, (const T*)x, (YT*)y, (long)n, thresh, inv_keep, (unsigned long long)seed, accumulate)
    if (x_dtype == MAGE_F32 && y_dtype == MAGE_F32) DROP(float, float);
    else if (x_dtype == MAGE_BF16 && y_dtype == MAGE_BF16) DROP(unsigned short, unsigned short);
    else if (x_dtype == MAGE_BF16 && y_dtype == MAGE_F32) DROP(unsigned short, float);
    else if (x_dtype == MAGE_F32 && y_dtype == MAGE_BF16) DROP(float, unsigned short);
    else { mage_set_error("mage_dropout: bad dtypes %d %d", x_dtype, y_dtype); return MAGE_EINVAL; }
#undef DROP
    MAGE_CHECK_LAUNCH("mage_dropout");
    return MAGE_OK;
}

extern "C" int mage_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                         int32_t step, float grad_scale, void* stream) {
    MAGE_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "mage_adam: bad arguments");
    MAGE_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "mage_adam: arenas must be 16-byte aligned");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr, beta1,
                       beta2, eps, bc1, bc2_sqrt, grad_scale);
    MAGE_CHECK_LAUNCH("mage_adam");
    return MAGE_OK;
}

// =================================================================================================================================
// Backward of the randomness branch of MAGE.forward (mage_model.py:601-609): GroupNorm of the Conv3d video prior (:264-297), the
// instance norm of ADAIN2D (:299-314), the reparameterisation + KL term (:569-573, :623).
// =================================================================================================================================
namespace {

// d act(t) / dt * dy;  act: 0 none, 1 ReLU, 2 SiLU
__device__ __forceinline__ float gn_dact(int act, float t, float dy) {
    if (act == 1) return t > 0.f ? dy : 0.f;
    if (act == 2) {
        const float sg = 1.0f / (1.0f + expf(-t));
        return dy * sg * (1.0f + t * (1.0f - sg));
    }
    return dy;
}

// y = act(xhat * gamma + beta + residual), xhat = (x - mean_{b,g}) * rstd_{b,g}.  With g = dy * act'(.):
//   dgamma_c = sum_{b,r} g xhat,  dbeta_c = sum_{b,r} g,  dx = rstd (gamma g - m1 - xhat m2),  m1 = mean_{(b,g)}(gamma g), m2 = mean(gamma g xhat).
// Pass 1: one workgroup per (sample, group); thread = (channel of the group, row phase); writes the per-(sample, channel) sums
// (fixed order) and red[b][g] = (m1, m2).  Row maps: x rows b*xs + xo + r, dy rows b*ds + dof + r, residual packed b*rows + r.
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ x, long xs, long xo, int rows_per_sample, int C, int groups,
                                                            const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ residual, int act,
                                                            const float* __restrict__ dy, long ds, long dof, float* __restrict__ red,
                                                            float* __restrict__ dgam_part, float* __restrict__ dbet_part) {
    __shared__ double sm[2][256];
    const int b = blockIdx.x, g = blockIdx.y, cpg = C / groups;
    const int cl = threadIdx.x % cpg, ph = threadIdx.x / cpg, nph = 256 / cpg;
    const int c = g * cpg + cl;
    const float mean = stats[((long)b * groups + g) * 2], rstd = stats[((long)b * groups + g) * 2 + 1];
    const float gm = gamma[c], bt = beta[c];
    double sg = 0.0, sgx = 0.0;
    for (int r = ph; r < rows_per_sample; r += nph) {
        const float xh = (x[((long)b * xs + xo + r) * C + c] - mean) * rstd;
        float t = xh * gm + bt;
        if (residual) t += residual[((long)b * rows_per_sample + r) * C + c];
        const float ge = gn_dact(act, t, dy[((long)b * ds + dof + r) * C + c]);
        sg += (double)ge;
        sgx += (double)ge * (double)xh;
    }
    sm[0][threadIdx.x] = sg;
    sm[1][threadIdx.x] = sgx;
    __syncthreads();
    if (threadIdx.x < cpg) {
        double a = 0.0, q = 0.0;
        for (int p = 0; p < nph; ++p) {
            a += sm[0][p * cpg + threadIdx.x];
            q += sm[1][p * cpg + threadIdx.x];
        }
        dbet_part[(long)b * C + c] = (float)a;
        dgam_part[(long)b * C + c] = (float)q;
        sm[0][threadIdx.x] = a * (double)gm;
        sm[1][threadIdx.x] = q * (double)gm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, q = 0.0;
        for (int i = 0; i < cpg; ++i) {
            a += sm[0][i];
            q += sm[1][i];
        }
        const double n = (double)rows_per_sample * cpg;
        red[((long)b * groups + g) * 2] = (float)(a / n);
        red[((long)b * groups + g) * 2 + 1] = (float)(q / n);
    }
}

// Pass 2 (elementwise): dx rows in x's row map; dres (optional, packed) = g.
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, long xs, long xo, int rows_per_sample, int C, int groups,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ residual, int act,
                                                           const float* __restrict__ dy, long ds, long dof, const float* __restrict__ red,
                                                           float* __restrict__ dx, float* __restrict__ dres, long total) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    const long orow = i / C;
    const int c = (int)(i - orow * C);
    const long b = orow / rows_per_sample, r = orow - b * rows_per_sample;
    const int cpg = C / groups;
    const f32x4 v = *(const f32x4*)(x + ((b * xs + xo + r) * C + c));
    const f32x4 gy = *(const f32x4*)(dy + ((b * ds + dof + r) * C + c));
    const f32x4 gm = *(const f32x4*)(gamma + c), bt = *(const f32x4*)(beta + c);
    f32x4 res = f32x4{0.f, 0.f, 0.f, 0.f};
    if (residual) res = *(const f32x4*)(residual + i);
    f32x4 o, ge4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long sg = (b * groups + (c + e) / cpg) * 2;
        const float mean = stats[sg], rstd = stats[sg + 1];
        const float xh = (v[e] - mean) * rstd;
        const float ge = gn_dact(act, xh * gm[e] + bt[e] + res[e], gy[e]);
        ge4[e] = ge;
        o[e] = rstd * (gm[e] * ge - red[sg] - xh * red[sg + 1]);
    }
    *(f32x4*)(dx + ((b * xs + xo + r) * C + c)) = o;
    if (dres) *(f32x4*)(dres + i) = ge4;
}

// out = gamma_map * IN(x) + beta_map over the P positions of (b, c) (adain_kernel in norm_attn.hip).  Given dout:
//   dgamma_map = dout * xhat, dbeta_map = dout (no kernel), dx = rstd (g - mean_p(g) - xhat mean_p(g xhat)), g = dout * gamma_map.
// grid = (B, C/64); block 256 = 64 channels x 4 position phases.
__global__ __launch_bounds__(256) void adain_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gmap, const float* __restrict__ dout,
                                                        float* __restrict__ dx, float* __restrict__ dgmap, int P, int C, float eps) {
    __shared__ float red[4][4][64];
    const int b = blockIdx.x, cl = threadIdx.x & 63, c = blockIdx.y * 64 + cl, ph = threadIdx.x >> 6;
    const long base = (long)b * P * C + c;
    float s = 0.f;
    for (int p = ph; p < P; p += 4) s += x[base + (long)p * C];
    red[0][ph][cl] = s;
    __syncthreads();
    const float mean = (red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl]) / (float)P;
    float q = 0.f;
    for (int p = ph; p < P; p += 4) {
        const float dlt = x[base + (long)p * C] - mean;
        q += dlt * dlt;
    }
    red[1][ph][cl] = q;
    __syncthreads();
    const float var = (red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl]) / (float)P;
    const float rstd = 1.0f / sqrtf(var + eps);
    float s1 = 0.f, s2 = 0.f;
    for (int p = ph; p < P; p += 4) {
        const long i = base + (long)p * C;
        const float xh = (x[i] - mean) * rstd, g = dout[i] * gmap[i];
        s1 += g;
        s2 += g * xh;
    }
    red[2][ph][cl] = s1;
    red[3][ph][cl] = s2;
    __syncthreads();
    const float m1 = (red[2][0][cl] + red[2][1][cl] + red[2][2][cl] + red[2][3][cl]) / (float)P;
    const float m2 = (red[3][0][cl] + red[3][1][cl] + red[3][2][cl] + red[3][3][cl]) / (float)P;
    for (int p = ph; p < P; p += 4) {
        const long i = base + (long)p * C;
        const float xh = (x[i] - mean) * rstd, go = dout[i];
        dgmap[i] = go * xh;
        dx[i] = rstd * (go * gmap[i] - m1 - xh * m2);
    }
}

// z = eps exp(logvar / 2) + mu; kl = -1/2 mean_b sum(1 + logvar - mu^2 - exp(logvar)).  With dz and c = dL/dkl / B:
//   dmu = dz + c mu,  dlogvar = dz eps exp(logvar / 2) / 2 - c (1 - exp(logvar)) / 2
__global__ void reparam_kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar, const float* __restrict__ eps,
                                      const float* __restrict__ dz, const float* __restrict__ coef, float* __restrict__ dmu,
                                      float* __restrict__ dlogvar, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c = coef[0], lv = logvar[i];
    dmu[i] = dz[i] + c * mu[i];
    dlogvar[i] = 0.5f * dz[i] * eps[i] * expf(0.5f * lv) - 0.5f * c * (1.0f - expf(lv));
}

}  // namespace

extern "C" int mage_groupnorm_bwd(const float* x, int64_t sample_stride_rows, int64_t row_off, int32_t n_samples, int32_t rows_per_sample,
                                  int32_t C, int32_t groups, const float* stats, const float* gamma, const float* beta, const float* residual,
                                  int32_t act, const float* dy, int64_t dy_sample_stride_rows, int64_t dy_row_off, float* red, float* dx,
                                  float* dres, float* dgamma_part, float* dbeta_part, void* stream) {
    MAGE_CHECK_ARG(x && stats && gamma && beta && dy && red && dx && dgamma_part && dbeta_part, "mage_groupnorm_bwd: null pointer");
    MAGE_CHECK_ARG(n_samples > 0 && rows_per_sample > 0 && groups > 0 && C % groups == 0 && C % 4 == 0 && (C / groups) <= 256 &&
                       256 % (C / groups) == 0 && act >= 0 && act <= 2,
                   "mage_groupnorm_bwd: C=%d groups=%d act=%d unsupported (channels per group must divide 256)", C, groups, act);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(n_samples, groups), dim3(256), 0, s, x, (long)sample_stride_rows, (long)row_off, rows_per_sample,
                       C, groups, stats, gamma, beta, residual, act, dy, (long)dy_sample_stride_rows, (long)dy_row_off, red, dgamma_part,
                       dbeta_part);
    const long total = (long)n_samples * rows_per_sample * C;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, x, (long)sample_stride_rows,
                       (long)row_off, rows_per_sample, C, groups, stats, gamma, beta, residual, act, dy, (long)dy_sample_stride_rows,
                       (long)dy_row_off, red, dx, dres, total);
    MAGE_CHECK_LAUNCH("mage_groupnorm_bwd");
    return MAGE_OK;
}

extern "C" int mage_adain_bwd(const float* x, const float* gamma_map, const float* dout, float* dx, float* dgamma_map, int32_t B, int32_t P,
                              int32_t C, float eps, void* stream) {
    MAGE_CHECK_ARG(x && gamma_map && dout && dx && dgamma_map && B > 0 && P > 0 && C > 0 && C % 64 == 0, "mage_adain_bwd: bad arguments");
    hipLaunchKernelGGL(adain_bwd_kernel, dim3(B, C / 64), dim3(256), 0, (hipStream_t)stream, x, gamma_map, dout, dx, dgamma_map, P, C, eps);
    MAGE_CHECK_LAUNCH("mage_adain_bwd");
    return MAGE_OK;
}

extern "C" int mage_reparam_kl_bwd(const float* mu, const float* logvar, const float* eps, const float* dz, const float* coef, float* dmu,
                                   float* dlogvar, int64_t n, void* stream) {
    MAGE_CHECK_ARG(mu && logvar && eps && dz && coef && dmu && dlogvar && n > 0, "mage_reparam_kl_bwd: bad arguments");
    hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps, dz, coef, dmu,
                       dlogvar, (long)n);
    MAGE_CHECK_LAUNCH("mage_reparam_kl_bwd");
    return MAGE_OK;
}

// d mean((a - b)^2) / da = 2 (a - b) / (rows * cols) * gout for the first `cols` columns of each row of a (row stride lda), zero for the
// padding columns up to ld_da (F.mse_loss of the MAGE+ latent prediction, mage_model.py:620).
namespace {
__global__ void mse_bwd_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb, long rows, int cols,
                               const float* __restrict__ gout, float* __restrict__ da, long ld_da, float inv_n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ld_da) return;
    const long r = i / ld_da;
    const int c = (int)(i - r * ld_da);
    da[i] = c < cols ? 2.0f * (a[r * lda + c] - b[r * ldb + c]) * inv_n * gout[0] : 0.f;
}
}  // namespace

extern "C" int mage_mse_bwd(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int32_t cols, const float* gout, float* da,
                            int64_t ld_da, void* stream) {
    MAGE_CHECK_ARG(a && b && gout && da && rows > 0 && cols > 0 && ld_da >= cols, "mage_mse_bwd: bad arguments");
    const long n = rows * ld_da;
    hipLaunchKernelGGL(mse_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, (long)lda, b, (long)ldb, (long)rows,
                       cols, gout, da, (long)ld_da, (float)(1.0 / ((double)rows * cols)));
    MAGE_CHECK_LAUNCH("mage_mse_bwd");
    return MAGE_OK;
}

// Backward of MaxPool2d(2) and of Upsample(scale 2, nearest) of the f8 VQ-VAE (vqvae_model.py:194-210), channels-last fp32.
namespace {
// dx of the 2x2 window = dy at the FIRST maximum in scan order (PyTorch's tie rule), zero elsewhere; one thread per output pixel x 4 channels.
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int N,
                                                           int H, int W, int C) {
    const int cq = C / 4, OH = H / 2, OW = W / 2;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)N * OH * OW * cq) return;
    const int c = (int)(gid % cq) * 4;
    const long pix = gid / cq;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((long)OW * OH));
    const long base = (((long)n * H + oy * 2) * W + ox * 2) * C + c;
    const long off[4] = {0, (long)C, (long)W * C, (long)W * C + C};
    f32x4 v[4], o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = *(const f32x4*)(x + base + off[k]);
        o[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 g = *(const f32x4*)(dy + pix * C + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int best = 0;
        float m = v[0][e];
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (v[k][e] > m) {
                m = v[k][e];
                best = k;
            }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k == best) o[k][e] = g[e];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *(f32x4*)(dx + base + off[k]) = o[k];
}

// dx[n, y, x] = sum of the 2x2 block of dy it was copied to.
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C) {
    const int cq = C / 4;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)N * H * W * cq) return;
    const int c = (int)(gid % cq) * 4;
    const long pix = gid / cq;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
    const long base = (((long)n * 2 * H + iy * 2) * 2 * W + ix * 2) * C + c;
    const f32x4 s = (*(const f32x4*)(dy + base) + *(const f32x4*)(dy + base + C)) +
                    (*(const f32x4*)(dy + base + (long)2 * W * C) + *(const f32x4*)(dy + base + (long)2 * W * C + C));
    *(f32x4*)(dx + pix * C + c) = s;
}
}  // namespace

extern "C" int mage_maxpool2_bwd(const float* x, const float* dy, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    MAGE_CHECK_ARG(x && dy && dx && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "mage_maxpool2_bwd: bad arguments");
    const long items = (long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, N, H, W, C);
    MAGE_CHECK_LAUNCH("mage_maxpool2_bwd");
    return MAGE_OK;
}

extern "C" int mage_upsample2_bwd(const float* dy, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    MAGE_CHECK_ARG(dy && dx && N > 0 && H > 0 && W > 0 && C % 4 == 0, "mage_upsample2_bwd: bad arguments");
    const long items = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(upsample2_bwd_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, H, W, C);
    MAGE_CHECK_LAUNCH("mage_upsample2_bwd");
    return MAGE_OK;
}

// y = r + dropout(x): the residual add of x + dropout(Linear(.)) (mage_model.py:48,52) without first copying r into y.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void dropout_add_kernel(const T* __restrict__ x, const float* __restrict__ r, float* __restrict__ y, long n,
                                                          unsigned thresh, float inv_keep, unsigned long long seed,
                                                          unsigned short* __restrict__ yb) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const f32x4 v = load4(x + i);
    f32x4 o = *(const f32x4*)(r + i);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (hash32(seed * 0x9e3779b97f4a7c15ULL + (unsigned long long)(i + e)) >= thresh) o[e] += v[e] * inv_keep;
    *(f32x4*)(y + i) = o;
    if (yb) store4(yb + i, o);
}
}  // namespace

// y = r + dropout(x) and yn = LayerNorm(y) in one pass (one wave per row, the arithmetic of layernorm_kernel on the registers that hold
// the new row): the residual add of a transformer block followed by the norm that opens the next branch (mage_model.py:48-52).
namespace {
template <typename T, typename OT, int VPL>
__global__ __launch_bounds__(256) void dropout_add_ln_kernel(const T* __restrict__ x, const float* __restrict__ r, float* __restrict__ y,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             OT* __restrict__ yn, long rows, int C, float eps, unsigned thresh,
                                                             float inv_keep, unsigned long long seedmul) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int c = j * 256 + lane * 4;
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < C) {
            const long i = row * C + c;
            const f32x4 b = load4(x + i);
            v[j] = *(const f32x4*)(r + i);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (hash32(seedmul + (unsigned long long)(i + e)) >= thresh) v[j][e] += b[e] * inv_keep;
            *(f32x4*)(y + i) = v[j];
        }
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int c = j * 256 + lane * 4;
        if (c < C) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dlt = v[j][e] - mean;
                q += dlt * dlt;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    OT* yr = yn + row * C;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int c = j * 256 + lane * 4;
        if (c < C) {
            const f32x4 gm = *(const f32x4*)(gamma + c), bt = *(const f32x4*)(beta + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * gm[e] + bt[e];
            store4(yr + c, o);
        }
    }
}

template <typename T, typename OT>
int dropout_add_ln_launch(const void* x, const float* r, float* y, const float* gamma, const float* beta, void* yn, int64_t rows, int32_t C,
                          float eps, float p, uint64_t seed, hipStream_t s) {
    const unsigned thresh = (unsigned)((double)p * 4294967296.0);
    const float inv_keep = 1.0f / (1.0f - p);
    const unsigned long long seedmul = (unsigned long long)seed * 0x9e3779b97f4a7c15ULL;
    const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
#define DAL(V) hipLaunchKernelGGL((dropout_add_ln_kernel<T, OT, V>), grid, blk, 0, s, (const T*)x, r, y, gamma, beta, (OT*)yn, (long)rows, C, eps, \
                                  thresh, inv_keep, seedmul)
    if (C <= 256) DAL(1);
    else if (C <= 512) DAL(2);
    else if (C <= 1024) DAL(4);
    else DAL(8);
#undef DAL
    MAGE_CHECK_LAUNCH("mage_dropout_add_layernorm");
    return MAGE_OK;
}
}  // namespace

extern "C" int mage_dropout_add_layernorm(const void* x, int32_t x_dtype, const float* r, float* y, const float* gamma, const float* beta,
                                          void* yn, int32_t yn_dtype, int64_t rows, int32_t C, float eps, float p, uint64_t seed,
                                          void* stream) {
    MAGE_CHECK_ARG(x && r && y && gamma && beta && yn && rows > 0 && C > 0 && C % 4 == 0 && C <= 2048 && p >= 0.f && p < 1.f,
                   "mage_dropout_add_layernorm: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == MAGE_F32 && yn_dtype == MAGE_F32) return dropout_add_ln_launch<float, float>(x, r, y, gamma, beta, yn, rows, C, eps, p, seed, s);
    if (x_dtype == MAGE_F32 && yn_dtype == MAGE_BF16) return dropout_add_ln_launch<float, unsigned short>(x, r, y, gamma, beta, yn, rows, C, eps, p, seed, s);
    if (x_dtype == MAGE_BF16 && yn_dtype == MAGE_BF16) return dropout_add_ln_launch<unsigned short, unsigned short>(x, r, y, gamma, beta, yn, rows, C, eps, p, seed, s);
    mage_set_error("mage_dropout_add_layernorm: bad dtypes %d %d", x_dtype, yn_dtype);
    return MAGE_EINVAL;
}

extern "C" int mage_dropout_add(const void* x, int32_t x_dtype, const float* r, float* y, void* y_bf16, int64_t n, float p, uint64_t seed,
                                void* stream) {
    MAGE_CHECK_ARG(x && r && y && n > 0 && n % 4 == 0 && p >= 0.f && p < 1.f && ((uintptr_t)y_bf16 & 7) == 0, "mage_dropout_add: bad arguments");
    const unsigned thresh = (unsigned)((double)p * 4294967296.0);
    const float inv_keep = 1.0f / (1.0f - p);
    const dim3 grid((unsigned)((n / 4 + 255) / 256)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == MAGE_F32) hipLaunchKernelGGL((dropout_add_kernel<float>), grid, blk, 0, s, (const float*)x, r, y, (long)n, thresh, inv_keep, (unsigned long long)seed, (unsigned short*)y_bf16);
    else if (x_dtype == MAGE_BF16) hipLaunchKernelGGL((dropout_add_kernel<unsigned short>), grid, blk, 0, s, (const unsigned short*)x, r, y, (long)n, thresh, inv_keep, (unsigned long long)seed, (unsigned short*)y_bf16);
    else { mage_set_error("mage_dropout_add: bad dtype %d", x_dtype); return MAGE_EINVAL; }
    MAGE_CHECK_LAUNCH("mage_dropout_add");
    return MAGE_OK;
}
