// LayerNorm, short-sequence multi-head attention (axial / text / cross), ADAIN, speed-embedding add.
// All HBM-bound: one pass over the data, 16-byte vector accesses, fp32 arithmetic.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

// ------------------------------------------------------------------------------------ LayerNorm
// One wave per row; two-pass (mean, then centred variance) entirely in registers.
template <typename OT, int VPL>   // VPL = float4 vectors per lane (C <= 256*VPL)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, OT* __restrict__ y, long rows,
                                                        int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    f32x4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int c = j * 256 + lane * 4;
        v[j] = (c < C) ? *(const f32x4*)(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
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
    OT* yr = y + row * C;
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

// ------------------------------------------------------------------------------------ attention
// Workgroup = (sequence, head group).  K and V of the head group are staged once in LDS in their storage type
// (bf16 stays bf16: half the LDS bytes and half the ds_read_b128 per key, twice the resident workgroups);
// each thread then owns (query i, head h) pairs: q in registers, scores in registers (NK_MAX <= 64), softmax in
// registers, output accumulated in registers, all fp32.  Lanes of a wave that share h read the same LDS address
// (broadcast); the (up to 4) distinct heads of a wave sit 64/128 B apart: conflict-free b128 reads.
__device__ __forceinline__ void load8f(const float* p, f32x4& a, f32x4& b) {
    a = *(const f32x4*)p;
    b = *(const f32x4*)(p + 4);
}
__device__ __forceinline__ void load8f(const unsigned short* p, f32x4& a, f32x4& b) {
    const uint4 r = *(const uint4*)p;
    a = f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
    b = f32x4{__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u), __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
}
__device__ __forceinline__ void copy8(const float* src, float* dst) {
    *(f32x4*)dst = *(const f32x4*)src;
    *(f32x4*)(dst + 4) = *(const f32x4*)(src + 4);
}
__device__ __forceinline__ void copy8(const unsigned short* src, unsigned short* dst) { *(uint4*)dst = *(const uint4*)src; }
__device__ __forceinline__ void load8f(const f16_t* p, f32x4& a, f32x4& b) {
    const uint4 r = *(const uint4*)p;
    a = widen4<f16_t>(uint2{r.x, r.y});
    b = widen4<f16_t>(uint2{r.z, r.w});
}
__device__ __forceinline__ void copy8(const f16_t* src, f16_t* dst) { *(uint4*)dst = *(const uint4*)src; }

template <typename T, int NK_MAX, typename OT = T>   // OT: T, or (T = float) split_bf16 / split_f16: the output leaves as split rows (common.h)
__global__ __launch_bounds__(256) void attention_kernel(const mage_attn_desc d, int hg) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* ks = (T*)smem_raw;                            // [nk][hg*32]
    const int rowf = hg * 32;
    T* vs = ks + (long)d.nk * rowf;
    const int s = blockIdx.x;
    const int h0 = blockIdx.y * hg;
    const int outer = s / d.inner, in = s - outer * d.inner;
    const long q_base = (long)outer * d.q_outer_stride + in;
    const long o_base = d.o_axis_stride ? (long)outer * d.o_outer_stride + in : q_base;      // out: q's row map unless the descriptor gives its own
    const long o_step = d.o_axis_stride ? d.o_axis_stride : d.q_axis_stride;
    const long kv_base = (long)outer * d.kv_outer_stride + in;
    const T* kp = (const T*)d.k;
    const T* vp = (const T*)d.v;
    // stage K, V: nk rows x (hg*32) columns, 8 elements per thread-step
    const int vec_per_row = rowf / 8;
    for (int e = threadIdx.x; e < d.nk * vec_per_row; e += 256) {
        const int j = e / vec_per_row, c = (e - j * vec_per_row) * 8;
        const long row = kv_base + (long)j * d.kv_axis_stride;
        copy8(kp + row * d.ldk + h0 * 32 + c, ks + j * rowf + c);
        copy8(vp + row * d.ldv + h0 * 32 + c, vs + j * rowf + c);
    }
    __syncthreads();
    int klen = d.nk;
    if (d.kv_len) klen = min(klen, d.kv_len[s / d.kv_len_div]);
    const unsigned drop_thresh = d.drop_p > 0.f ? (unsigned)((double)d.drop_p * 4294967296.0) : 0u;
    const float inv_keep = 1.0f / (1.0f - d.drop_p);
    const T* qp = (const T*)d.q;
    OT* op = (OT*)d.out;
    const int ldo = std::is_same<OT, T>::value ? d.ldo : d.ldo >> 1;   // split rows: ldo counts 16-bit elements, a logical element is 4 bytes
    for (int p = threadIdx.x; p < d.nq * hg; p += 256) {
        const int hl = p / d.nq, i = p - hl * d.nq;
        const long row = q_base + (long)i * d.q_axis_stride;
        f32x4 q[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            load8f(qp + row * d.ldq + (h0 + hl) * 32 + c * 8, q[2 * c], q[2 * c + 1]);
            q[2 * c] *= d.scale;
            q[2 * c + 1] *= d.scale;
        }
        const int jmax = d.causal ? min(klen, i + 1 + (d.nk - d.nq)) : klen;   // bottom-right aligned causal mask
        float sc[NK_MAX];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NK_MAX; ++j) {
            float a = -INFINITY;
            if (j < jmax) {
                const T* kr = ks + j * rowf + hl * 32;
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 k0, k1;
                    load8f(kr + c * 8, k0, k1);
                    a0 += q[2 * c][0] * k0[0] + q[2 * c][1] * k0[1] + q[2 * c][2] * k0[2] + q[2 * c][3] * k0[3];
                    a1 += q[2 * c + 1][0] * k1[0] + q[2 * c + 1][1] * k1[1] + q[2 * c + 1][2] * k1[2] + q[2 * c + 1][3] * k1[3];
                }
                a = a0 + a1;
            }
            sc[j] = a;
            mx = fmaxf(mx, a);
        }
        float den = 0.f;
        f32x4 o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NK_MAX; ++j) {
            if (j < jmax) {
                float pj = expf(sc[j] - mx);
                den += pj;
                if (drop_thresh) {                   // dropout on the probabilities (train() of the text encoder): the sum stays unmasked
                    const unsigned long long idx = (((unsigned long long)s * d.n_head + (h0 + hl)) * d.nq + i) * d.nk + j;
                    pj = hash32(d.drop_seed * 0x9e3779b97f4a7c15ULL + idx) >= drop_thresh ? pj * inv_keep : 0.f;
                }
                const T* vr = vs + j * rowf + hl * 32;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 v0, v1;
                    load8f(vr + c * 8, v0, v1);
                    o[2 * c] += pj * v0;
                    o[2 * c + 1] += pj * v1;
                }
            }
        }
        const float inv = 1.0f / den;
        OT* orow = op + (o_base + (long)i * o_step) * ldo + (h0 + hl) * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) store8(orow + c * 8, o[2 * c] * inv, o[2 * c + 1] * inv);
    }
}

// ------------------------------------------------------------------------------------ attention on the matrix cores
// bf16, nq <= 32 and nk <= 32 (every axial attention of the decoder, frames_length up to 32): one wave per (sequence, head),
// three MFMA steps per block of 16 queries (and per block of 16 keys).
//   S^T = K Q^T   (v_mfma_f32_16x16x32_bf16, operands straight from global: lane (r = lane&15, g = lane>>4) loads the 16
//                  bytes [g*8, g*8+8) of row r of K resp. Q): the result lane (i = lane&15, g) holds S^T[j = 4g+e][i], e=0..3,
//                  i.e. four consecutive keys of ITS query -- exactly the B-operand layout of the next MFMA (swapped-QK idiom).
//   softmax over j: 4 values in the lane, then across the 4 lane groups (two xor-shuffles).
//   O^T = V^T P^T (v_mfma_f32_16x16x16_bf16 per 16-dim half; P split into bf16 hi + lo so that P V keeps fp32-class weights;
//                  A operand = V^T: lane (n = lane&15, g) needs V[j = 4g+e][n]: four 2-byte reads of the V rows staged in LDS).
// Only V goes through LDS (16.5 KB per workgroup: 9 workgroups per CU keep HBM busy); the thread-per-(query, head) kernel
// below converts every K/V element to fp32 once per query (VALU time ~ HBM time at bf16); here the arithmetic is ~40
// vector instructions per (sequence, head).  Output: a lane swap gives every lane 8 consecutive dims -> one 16-byte store.
typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8;
typedef __attribute__((ext_vector_type(4))) short ashort4;
typedef __attribute__((ext_vector_type(4))) _Float16 ahalf4;
// HT = the 16-bit element type of q, k, v and out: unsigned short (bf16) or f16_t (MAGE_F16).  P = hi + lo in that type -- bf16: hi = the
// truncation of p, lo = the truncated residue (8 + 8 bits); f16: hi = round(p), lo = round(p - hi) (11 + 11 bits; p <= 1, the residue is
// >= 2^-24 p: inside the f16 subnormal grid's 6e-8) -- so that P V keeps fp32-class weights in both.
template <typename HT> __device__ __forceinline__ void attn_p_split(float p, short& hi, short& lo) {
    if constexpr (std::is_same<HT, f16_t>::value) {
        const _Float16 h = (_Float16)p;
        hi = __builtin_bit_cast(short, h);
        lo = __builtin_bit_cast(short, (_Float16)(p - (float)h));
    } else {
        const unsigned hb = __float_as_uint(p) & 0xffff0000u;
        hi = (short)(hb >> 16);
        lo = (short)(__float_as_uint(p - __uint_as_float(hb)) >> 16);
    }
}
template <typename HT> __device__ __forceinline__ f32x4 attn_mfma_pv(const ashort4& vt, const ashort4& p, const f32x4& o) {
    if constexpr (std::is_same<HT, f16_t>::value)
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(ahalf4, vt), __builtin_bit_cast(ahalf4, p), o, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vt, p, o, 0, 0, 0);
}
template <typename HT> __device__ __forceinline__ f32x4 attn_mfma_qk(const uint4& k, const uint4& q) {
    return mfma16x16x32<HT>(__builtin_bit_cast(u32x4, k), __builtin_bit_cast(u32x4, q), f32x4{0.f, 0.f, 0.f, 0.f});
}

// NKB = key blocks of 16 (nk <= 16*NKB); queries are walked in blocks of 16 (nq <= 32 at the call sites: frames_length 32).
// MAXH = heads per wave (ceil(n_head / 4) rounded up to 2, 4 or 8): sizes the preloaded fragment registers; 16 heads -> 4, which
// keeps the kernel at a register count that lets 4-5 workgroups share a CU (a memory-bound kernel lives off that).
template <int NKB, int MAXH, typename HT = unsigned short>
__global__ __launch_bounds__(256) void attention_mfma_kernel(const mage_attn_desc d) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned short* vs = (unsigned short*)smem_raw;     // [16*NKB][n_head*32 + 16]: rows >= nk are zero
    const int rowf = d.n_head * 32, vpitch = rowf + 16;
    const int s = blockIdx.x;
    const int outer = s / d.inner, in = s - outer * d.inner;
    const long q_base = (long)outer * d.q_outer_stride + in;
    const long o_base = d.o_axis_stride ? (long)outer * d.o_outer_stride + in : q_base;      // out: q's row map unless the descriptor gives its own
    const long o_step = d.o_axis_stride ? d.o_axis_stride : d.q_axis_stride;
    const long kv_base = (long)outer * d.kv_outer_stride + in;
    const unsigned short* qp = (const unsigned short*)d.q;
    const unsigned short* kp = (const unsigned short*)d.k;
    const unsigned short* vp = (const unsigned short*)d.v;
    HT* op = (HT*)d.out;
    const int vec_per_row = rowf / 8;
    for (int e = threadIdx.x; e < 16 * NKB * vec_per_row; e += 256) {
        const int j = e / vec_per_row, c = (e - j * vec_per_row) * 8;
        uint4 val = uint4{0u, 0u, 0u, 0u};
        if (j < d.nk) val = *(const uint4*)(vp + (kv_base + (long)j * d.kv_axis_stride) * d.ldv + c);
        *(uint4*)(vs + j * vpitch + c) = val;
    }
    int klen = d.nk;
    if (d.kv_len) klen = min(klen, d.kv_len[s / d.kv_len_div]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    // all of this wave's K fragments up front (independent 16-byte loads; out-of-range keys clamped, masked below)
    uint4 kf[MAXH][NKB];
#pragma unroll
    for (int t = 0; t < MAXH; ++t) {
        const int h = wave + 4 * t;
        if (h < d.n_head) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const long krow = kv_base + (long)min(kb * 16 + r, d.nk - 1) * d.kv_axis_stride;
                kf[t][kb] = *(const uint4*)(kp + krow * d.ldk + h * 32 + g * 8);
            }
        }
    }
    __syncthreads();
    const int nqb = (d.nq + 15) >> 4;
    for (int qb = 0; qb < nqb; ++qb) {
        const int qi = qb * 16 + r;                                           // this lane's query
        const long qrow = q_base + (long)min(qi, d.nq - 1) * d.q_axis_stride;
        const int jmax = d.causal ? min(klen, qi + 1 + (d.nk - d.nq)) : klen;  // query qi sees keys j < jmax
        uint4 qf[MAXH];
#pragma unroll
        for (int t = 0; t < MAXH; ++t) {
            const int h = wave + 4 * t;
            if (h < d.n_head) qf[t] = *(const uint4*)(qp + qrow * d.ldq + h * 32 + g * 8);
        }
#pragma unroll
        for (int t = 0; t < MAXH; ++t) {
            const int h = wave + 4 * t;
            if (h >= d.n_head) break;
            f32x4 st[NKB];
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                st[kb] = attn_mfma_qk<HT>(kf[t][kb], qf[t]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st[kb][e] = (kb * 16 + 4 * g + e < jmax) ? st[kb][e] * d.scale : -INFINITY;
                    mx = fmaxf(mx, st[kb][e]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float den = 0.f;
            ashort4 phi[NKB], plo[NKB];                                       // P = hi + lo in bf16 (lo = the truncation residue)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = (kb * 16 + 4 * g + e < jmax) ? expf(st[kb][e] - mx) : 0.f;
                    den += p;
                    short ph, pl;
                    attn_p_split<HT>(p, ph, pl);
                    phi[kb][e] = ph;
                    plo[kb][e] = pl;
                }
            den += __shfl_xor(den, 16);
            den += __shfl_xor(den, 32);
            f32x4 o[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                o[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    const unsigned short* vr = vs + (kb * 16 + 4 * g) * vpitch + h * 32 + b * 16 + r;
                    ashort4 vt;
#pragma unroll
                    for (int e = 0; e < 4; ++e) vt[e] = (short)vr[e * vpitch];
                    o[b] = attn_mfma_pv<HT>(vt, phi[kb], o[b]);
                    o[b] = attn_mfma_pv<HT>(vt, plo[kb], o[b]);
                }
            }
            // lane (i = r, g) holds O[i][b*16 + 4g + e]; swap halves between neighbouring lane groups: 8 consecutive dims per lane
            const float inv = 1.0f / den;
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(o[0][e]), __float_as_uint(o[1][e]), false, false);
                v0[e] = __uint_as_float(sw[0]) * inv;
                v1[e] = __uint_as_float(sw[1]) * inv;
            }
            if (qi < d.nq) {
                const int col = h * 32 + 16 * (g & 1) + 8 * (g >> 1);
                store8(op + (o_base + (long)qi * o_step) * d.ldo + col, v0, v1);
            }
        }
    }
}

// Few queries, many sequences (the incremental AR step's temporal attention: nq = 1 or 2 new positions against a K,V cache of up to
// 32, 16384 sequences at cfg2): one WAVE per sequence, no LDS, no barrier.  Per (sequence, head) the SAME instruction sequence on the
// same operand values as attention_mfma_kernel (S^T = K Q^T by one 16x16x32 MFMA, in-lane softmax + two xor-shuffles, P = hi + lo,
// O^T = V^T P^T by 16x16x16 MFMAs) -- the incremental loop's tokens stay bit-identical to the full pass's -- with the V rows in a
// wave-private LDS tile (rows beyond nk are clamped copies: their probabilities are exactly 0) and four heads' loads in flight before
// the first MFMA.  The workgroup-per-sequence kernel
// spends its time in per-workgroup latency at this shape (V staging + barrier for ONE query): 107 us per launch at cfg2 against
// ~55 us of K,V cache bytes.
// DPP row shifts (within each 16-lane row): row_shr<N>: lane r takes src of lane r - N (lanes r < N keep old); row_shl<N>: lane r takes
// lane r + N (lanes r >= 16 - N: themselves).  N = 0: src.
template <int N> __device__ __forceinline__ float row_shr_f(float old, float src) {
    if constexpr (N == 0) return src;
    else return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(old), __float_as_uint(src), 0x110 + N, 0xf, 0xf, false));
}
template <int N> __device__ __forceinline__ unsigned row_shl_u(unsigned src) {
    if constexpr (N == 0) return src;
    else return __builtin_amdgcn_update_dpp(src, src, 0x100 + N, 0xf, 0xf, false);
}
template <int N> __device__ __forceinline__ float row_shl_f(float src) { return __uint_as_float(row_shl_u<N>(__float_as_uint(src))); }
template <int N> __device__ __forceinline__ ashort4 row_shl_s4(ashort4 src) {
    const uint2 u = __builtin_bit_cast(uint2, src);
    return __builtin_bit_cast(ashort4, uint2{row_shl_u<N>(u.x), row_shl_u<N>(u.y)});
}

template <int NKB, typename HT = unsigned short>
__global__ __launch_bounds__(256) void attention_mfma_fewq_kernel(const mage_attn_desc d) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= d.n_seq) return;
    const int outer = s / d.inner, in = s - outer * d.inner;
    const long q_base = (long)outer * d.q_outer_stride + in;
    const long o_base = d.o_axis_stride ? (long)outer * d.o_outer_stride + in : q_base;      // out: q's row map unless the descriptor gives its own
    const long o_step = d.o_axis_stride ? d.o_axis_stride : d.q_axis_stride;
    const long kv_base = (long)outer * d.kv_outer_stride + in;
    const unsigned short* qp = (const unsigned short*)d.q;
    const unsigned short* kp = (const unsigned short*)d.k;
    const unsigned short* vp = (const unsigned short*)d.v;
    HT* op = (HT*)d.out;
    int klen = d.nk;
    if (d.kv_len) klen = min(klen, d.kv_len[s / d.kv_len_div]);
    const int r = lane & 15, g = lane >> 4;
    const int qi = r;                                                         // nq <= 16: one query block
    const long qrow = q_base + (long)min(qi, d.nq - 1) * d.q_axis_stride;
    long krow[NKB], krow_v[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        krow[kb] = (kv_base + (long)min(kb * 16 + r, d.nk - 1) * d.kv_axis_stride) * d.ldk;
        krow_v[kb] = (kv_base + (long)min(kb * 16 + r, d.nk - 1) * d.kv_axis_stride) * d.ldv;
    }
    // chunks of CH heads: their loads are all in flight before the first MFMA.  (Measured: a software-pipelined variant -- chunk c+1's
    // loads under chunk c's arithmetic -- needs 168 VGPRs, 2 waves per SIMD, and is SLOWER: 105 us at nk = 2 against 74; this kernel
    // lives off occupancy.)
    // V^T fragments (lane (n, g) needs V[4g+e][n], e = 0..3: a column walk): each lane loads 16 bytes of a V row like a K fragment,
    // parks them in a WAVE-PRIVATE LDS tile (no barrier: one wave's DS operations execute in order) and reads its four 2-byte values
    // back.  (Fetched as 2-byte global loads the kernel was bound by the texture addresser: 75 us at nk = 2.)
    constexpr int CH = 4, VP = 40;                                           // 80-byte LDS rows
    __shared__ unsigned short vsm[4][CH][NKB * 16][VP];
    const int wv = threadIdx.x >> 6;
    uint4 kf[1][CH][NKB], qf[1][CH];
    ashort4 vt[1][CH][2][NKB];
    auto load_chunk = [&](int buf, int h0) {
#pragma unroll
        for (int t = 0; t < CH; ++t) {
            const int h = min(h0 + t, d.n_head - 1);
            qf[buf][t] = *(const uint4*)(qp + qrow * d.ldq + h * 32 + g * 8);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                kf[buf][t][kb] = *(const uint4*)(kp + krow[kb] + h * 32 + g * 8);
                *(uint4*)&vsm[wv][t][kb * 16 + r][g * 8] = *(const uint4*)(vp + krow_v[kb] + h * 32 + g * 8);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < CH; ++t)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 4; ++e) vt[buf][t][b][kb][e] = (short)vsm[wv][t][kb * 16 + 4 * g + e][b * 16 + r];
        __builtin_amdgcn_wave_barrier();
    };
    // One softmax for the CH heads of a chunk.  With nq <= 2 only result columns (lanes l15 =) 0, 1 of a head's 16 x 16 score block
    // are queries; the other 14 repeat them.  Every VALU instruction costs a wave 4 cycles whatever its useful lanes, and the softmax
    // (exp, max / sum trees, the hi + lo split) is ~100 of a head's ~200 instructions: so the CH heads' two columns are moved side by
    // side (head t -> lanes 2t, 2t+1 of each 16-lane row: one DPP row shift per register), scaled / masked / exponentiated / summed ONCE,
    // and moved back as each head's P operand.  Per element the same operations in the same order as the one-head form (the xor-16 /
    // xor-32 partners are the same lanes of the other key groups): bit-identical rows.
    const int jmax_p = d.causal ? min(klen, (r & 1) + 1 + (d.nk - d.nq)) : klen;      // packed lanes: query = lane parity
    auto compute_chunk = [&](int buf, int h0) {
        f32x4 sp[NKB];                                                        // packed scores
        auto gather = [&](auto T) {
            constexpr int t = decltype(T)::value;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const f32x4 stt = attn_mfma_qk<HT>(kf[buf][t][kb], qf[buf][t]);
#pragma unroll
                for (int e = 0; e < 4; ++e) sp[kb][e] = row_shr_f<2 * t>(sp[kb][e], stt[e]);
            }
        };
        gather(std::integral_constant<int, 0>{});
        gather(std::integral_constant<int, 1>{});
        gather(std::integral_constant<int, 2>{});
        gather(std::integral_constant<int, 3>{});
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sp[kb][e] = (kb * 16 + 4 * g + e < jmax_p) ? sp[kb][e] * d.scale : -INFINITY;
                mx = fmaxf(mx, sp[kb][e]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float den = 0.f;
        ashort4 phi_p[NKB], plo_p[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = (kb * 16 + 4 * g + e < jmax_p) ? expf(sp[kb][e] - mx) : 0.f;
                den += p;
                short ph, pl;
                attn_p_split<HT>(p, ph, pl);
                phi_p[kb][e] = ph;
                plo_p[kb][e] = pl;
            }
        den += __shfl_xor(den, 16);
        den += __shfl_xor(den, 32);
        const float inv_p = 1.0f / den;
        auto finish = [&](auto T) {
            constexpr int t = decltype(T)::value;
            const int h = h0 + t;
            if (h >= d.n_head) return;
            ashort4 phi[NKB], plo[NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                phi[kb] = row_shl_s4<2 * t>(phi_p[kb]);
                plo[kb] = row_shl_s4<2 * t>(plo_p[kb]);
            }
            const float inv = row_shl_f<2 * t>(inv_p);
            f32x4 o[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                o[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    o[b] = attn_mfma_pv<HT>(vt[buf][t][b][kb], phi[kb], o[b]);
                    o[b] = attn_mfma_pv<HT>(vt[buf][t][b][kb], plo[kb], o[b]);
                }
            }
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(o[0][e]), __float_as_uint(o[1][e]), false, false);
                v0[e] = __uint_as_float(sw[0]) * inv;
                v1[e] = __uint_as_float(sw[1]) * inv;
            }
            if (qi < d.nq) {
                const int col = h * 32 + 16 * (g & 1) + 8 * (g >> 1);
                store8(op + (o_base + (long)qi * o_step) * d.ldo + col, v0, v1);
            }
        };
        finish(std::integral_constant<int, 0>{});
        finish(std::integral_constant<int, 1>{});
        finish(std::integral_constant<int, 2>{});
        finish(std::integral_constant<int, 3>{});
    };
    for (int h0 = 0; h0 < d.n_head; h0 += CH) {
        load_chunk(0, h0);
        compute_chunk(0, h0);
    }
}

// The fast parity mode's axial attention (q, k, v and out are SPLIT rows of f16 pieces, common.h): attention_mfma_kernel's structure
// with every product as three f16 MFMA passes -- S^T = (K_hi Q_lo^T + K_lo Q_hi^T) 2^-11 + K_hi Q_hi^T by v_mfma_f32_16x16x32_f16,
// softmax in fp32, P = hi + lo 2^-11 in f16 pieces, O^T = (V_hi^T P_lo^T + V_lo^T P_hi^T) 2^-11 + V_hi^T P_hi^T by 16x16x16 --
// 22-bit operands, fp32 accumulation: fp32-class like the split GEMMs, in place of the thread-per-(query, head) fp32 kernel (540 vs
// ~200 us per launch at cfg2).  Logical column c of a split row sits at 16-bit index (c / 64) * 128 + piece * 64 + c % 64.
__device__ __forceinline__ int split_col(int c, int piece) { return ((c >> 6) << 7) + (piece << 6) + (c & 63); }

template <int NKB, int MAXH>
__global__ __launch_bounds__(256) void attention_mfma_split_kernel(const mage_attn_desc d) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned short* vs = (unsigned short*)smem_raw;     // [16*NKB][2*n_head*32 + 16]: physical (split) V rows; rows >= nk are zero
    const int prow = d.n_head * 64, vpitch = prow + 16;
    const int s = blockIdx.x;
    const int outer = s / d.inner, in = s - outer * d.inner;
    const long q_base = (long)outer * d.q_outer_stride + in;
    const long o_base = d.o_axis_stride ? (long)outer * d.o_outer_stride + in : q_base;      // out: q's row map unless the descriptor gives its own
    const long o_step = d.o_axis_stride ? d.o_axis_stride : d.q_axis_stride;
    const long kv_base = (long)outer * d.kv_outer_stride + in;
    const unsigned short* qp = (const unsigned short*)d.q;
    const unsigned short* kp = (const unsigned short*)d.k;
    const unsigned short* vp = (const unsigned short*)d.v;
    split_f16* op = (split_f16*)d.out;
    const int ldo = d.ldo >> 1;
    const int vec_per_row = prow / 8;
    for (int e = threadIdx.x; e < 16 * NKB * vec_per_row; e += 256) {
        const int j = e / vec_per_row, c = (e - j * vec_per_row) * 8;
        uint4 val = uint4{0u, 0u, 0u, 0u};
        if (j < d.nk) val = *(const uint4*)(vp + (kv_base + (long)j * d.kv_axis_stride) * d.ldv + c);
        *(uint4*)(vs + j * vpitch + c) = val;
    }
    int klen = d.nk;
    if (d.kv_len) klen = min(klen, d.kv_len[s / d.kv_len_div]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    constexpr float LO = 1.0f / MAGE_F16_LO_SCALE;
    uint4 kf[MAXH][NKB][2];
#pragma unroll
    for (int t = 0; t < MAXH; ++t) {
        const int h = wave + 4 * t;
        if (h < d.n_head) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const long krow = (kv_base + (long)min(kb * 16 + r, d.nk - 1) * d.kv_axis_stride) * d.ldk;
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) kf[t][kb][pc] = *(const uint4*)(kp + krow + split_col(h * 32 + g * 8, pc));
            }
        }
    }
    __syncthreads();
    const int nqb = (d.nq + 15) >> 4;
    for (int qb = 0; qb < nqb; ++qb) {
        const int qi = qb * 16 + r;
        const long qrow = (q_base + (long)min(qi, d.nq - 1) * d.q_axis_stride) * d.ldq;
        const int jmax = d.causal ? min(klen, qi + 1 + (d.nk - d.nq)) : klen;
        uint4 qf[MAXH][2];
#pragma unroll
        for (int t = 0; t < MAXH; ++t) {
            const int h = wave + 4 * t;
            if (h < d.n_head) {
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) qf[t][pc] = *(const uint4*)(qp + qrow + split_col(h * 32 + g * 8, pc));
            }
        }
#pragma unroll
        for (int t = 0; t < MAXH; ++t) {
            const int h = wave + 4 * t;
            if (h >= d.n_head) break;
            f32x4 st[NKB];
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[t][kb][0]), __builtin_bit_cast(f16x8, qf[t][1]),
                                                                 f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[t][kb][1]), __builtin_bit_cast(f16x8, qf[t][0]), a, 0, 0, 0);
                a *= LO;
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[t][kb][0]), __builtin_bit_cast(f16x8, qf[t][0]), a, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = (kb * 16 + 4 * g + e < jmax) ? a[e] * d.scale : -INFINITY;
                    mx = fmaxf(mx, a[e]);
                }
                st[kb] = a;
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float den = 0.f;
            ahalf4 phi[NKB], plo[NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = (kb * 16 + 4 * g + e < jmax) ? expf(st[kb][e] - mx) : 0.f;
                    den += p;
                    const _Float16 ph = (_Float16)p;
                    phi[kb][e] = ph;
                    plo[kb][e] = (_Float16)((p - (float)ph) * MAGE_F16_LO_SCALE);
                }
            den += __shfl_xor(den, 16);
            den += __shfl_xor(den, 32);
            f32x4 o[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                ahalf4 vth[NKB], vtl[NKB];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    const unsigned short* vr = vs + (kb * 16 + 4 * g) * vpitch;
                    const int ch = split_col(h * 32 + b * 16 + r, 0), cl = split_col(h * 32 + b * 16 + r, 1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vth[kb][e] = __builtin_bit_cast(_Float16, vr[e * vpitch + ch]);
                        vtl[kb][e] = __builtin_bit_cast(_Float16, vr[e * vpitch + cl]);
                    }
                }
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    a = __builtin_amdgcn_mfma_f32_16x16x16f16(vth[kb], plo[kb], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x16f16(vtl[kb], phi[kb], a, 0, 0, 0);
                }
                a *= LO;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) a = __builtin_amdgcn_mfma_f32_16x16x16f16(vth[kb], phi[kb], a, 0, 0, 0);
                o[b] = a;
            }
            const float inv = 1.0f / den;
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(o[0][e]), __float_as_uint(o[1][e]), false, false);
                v0[e] = __uint_as_float(sw[0]) * inv;
                v1[e] = __uint_as_float(sw[1]) * inv;
            }
            if (qi < d.nq) {
                const int col = h * 32 + 16 * (g & 1) + 8 * (g >> 1);
                store8(op + (o_base + (long)qi * o_step) * ldo + col, v0, v1);
            }
        }
    }
}

// The few-query form of attention_mfma_split_kernel (the incremental step's temporal attention in f16x3 mode): wave per sequence, the V
// pieces in a wave-private LDS tile, per (sequence, head) the same instruction sequence as the workgroup-per-sequence kernel (the
// incremental loop's tokens stay bit-identical to the full pass's).
template <int NKB>
__global__ __launch_bounds__(256) void attention_mfma_split_fewq_kernel(const mage_attn_desc d) {
    constexpr int CH = 2, VP = 40;
    __shared__ unsigned short vsm[4][CH][2][NKB * 16][VP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = blockIdx.x * 4 + wv;
    if (s >= d.n_seq) return;
    const int outer = s / d.inner, in = s - outer * d.inner;
    const long q_base = (long)outer * d.q_outer_stride + in;
    const long o_base = d.o_axis_stride ? (long)outer * d.o_outer_stride + in : q_base;      // out: q's row map unless the descriptor gives its own
    const long o_step = d.o_axis_stride ? d.o_axis_stride : d.q_axis_stride;
    const long kv_base = (long)outer * d.kv_outer_stride + in;
    const unsigned short* qp = (const unsigned short*)d.q;
    const unsigned short* kp = (const unsigned short*)d.k;
    const unsigned short* vp = (const unsigned short*)d.v;
    split_f16* op = (split_f16*)d.out;
    const int ldo = d.ldo >> 1;
    int klen = d.nk;
    if (d.kv_len) klen = min(klen, d.kv_len[s / d.kv_len_div]);
    const int r = lane & 15, g = lane >> 4;
    const int qi = r;
    const long qrow = (q_base + (long)min(qi, d.nq - 1) * d.q_axis_stride) * d.ldq;
    const int jmax_p = d.causal ? min(klen, (r & 1) + 1 + (d.nk - d.nq)) : klen;      // packed lanes: query = lane parity
    constexpr float LO = 1.0f / MAGE_F16_LO_SCALE;
    long krow[NKB], vrow[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        krow[kb] = (kv_base + (long)min(kb * 16 + r, d.nk - 1) * d.kv_axis_stride) * d.ldk;
        vrow[kb] = (kv_base + (long)min(kb * 16 + r, d.nk - 1) * d.kv_axis_stride) * d.ldv;
    }
    for (int h0 = 0; h0 < d.n_head; h0 += CH) {
        uint4 kf[CH][NKB][2], qf[CH][2];
#pragma unroll
        for (int t = 0; t < CH; ++t) {
            const int h = min(h0 + t, d.n_head - 1);
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const int c = split_col(h * 32 + g * 8, pc);
                qf[t][pc] = *(const uint4*)(qp + qrow + c);
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    kf[t][kb][pc] = *(const uint4*)(kp + krow[kb] + c);
                    *(uint4*)&vsm[wv][t][pc][kb * 16 + r][g * 8] = *(const uint4*)(vp + vrow[kb] + c);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // one softmax for the chunk's CH = 2 heads (their query columns side by side: attention_mfma_fewq_kernel)
        f32x4 sp[NKB];
        auto gather = [&](auto T) {
            constexpr int t = decltype(T)::value;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[t][kb][0]), __builtin_bit_cast(f16x8, qf[t][1]),
                                                                 f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[t][kb][1]), __builtin_bit_cast(f16x8, qf[t][0]), a, 0, 0, 0);
                a *= LO;
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[t][kb][0]), __builtin_bit_cast(f16x8, qf[t][0]), a, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) sp[kb][e] = row_shr_f<2 * t>(sp[kb][e], a[e]);
            }
        };
        gather(std::integral_constant<int, 0>{});
        gather(std::integral_constant<int, 1>{});
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sp[kb][e] = (kb * 16 + 4 * g + e < jmax_p) ? sp[kb][e] * d.scale : -INFINITY;
                mx = fmaxf(mx, sp[kb][e]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float den = 0.f;
        ahalf4 phi_p[NKB], plo_p[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = (kb * 16 + 4 * g + e < jmax_p) ? expf(sp[kb][e] - mx) : 0.f;
                den += p;
                const _Float16 ph = (_Float16)p;
                phi_p[kb][e] = ph;
                plo_p[kb][e] = (_Float16)((p - (float)ph) * MAGE_F16_LO_SCALE);
            }
        den += __shfl_xor(den, 16);
        den += __shfl_xor(den, 32);
        const float inv_p = 1.0f / den;
        auto finish = [&](auto T) {
            constexpr int t = decltype(T)::value;
            const int h = h0 + t;
            if (h >= d.n_head) return;
            ahalf4 phi[NKB], plo[NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                phi[kb] = __builtin_bit_cast(ahalf4, row_shl_s4<2 * t>(__builtin_bit_cast(ashort4, phi_p[kb])));
                plo[kb] = __builtin_bit_cast(ahalf4, row_shl_s4<2 * t>(__builtin_bit_cast(ashort4, plo_p[kb])));
            }
            const float inv = row_shl_f<2 * t>(inv_p);
            f32x4 o[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                ahalf4 vth[NKB], vtl[NKB];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vth[kb][e] = __builtin_bit_cast(_Float16, vsm[wv][t][0][kb * 16 + 4 * g + e][b * 16 + r]);
                        vtl[kb][e] = __builtin_bit_cast(_Float16, vsm[wv][t][1][kb * 16 + 4 * g + e][b * 16 + r]);
                    }
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    a = __builtin_amdgcn_mfma_f32_16x16x16f16(vth[kb], plo[kb], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x16f16(vtl[kb], phi[kb], a, 0, 0, 0);
                }
                a *= LO;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) a = __builtin_amdgcn_mfma_f32_16x16x16f16(vth[kb], phi[kb], a, 0, 0, 0);
                o[b] = a;
            }
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(o[0][e]), __float_as_uint(o[1][e]), false, false);
                v0[e] = __uint_as_float(sw[0]) * inv;
                v1[e] = __uint_as_float(sw[1]) * inv;
            }
            if (qi < d.nq) {
                const int col = h * 32 + 16 * (g & 1) + 8 * (g >> 1);
                store8(op + (o_base + (long)qi * o_step) * ldo + col, v0, v1);
            }
        };
        finish(std::integral_constant<int, 0>{});
        finish(std::integral_constant<int, 1>{});
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------ ADAIN
// grid = (B, C/64); block 256 = 64 channels x 4 position phases.
__global__ __launch_bounds__(256) void adain_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ out, int P,
                                                    int C, float eps) {
    __shared__ float red[2][4][64];
    const int b = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    const long base = (long)b * P * C + c;
    float s = 0.f;
    for (int p = ph; p < P; p += 4) s += x[base + (long)p * C];
    red[0][ph][threadIdx.x & 63] = s;
    __syncthreads();
    const float mean = (red[0][0][threadIdx.x & 63] + red[0][1][threadIdx.x & 63] + red[0][2][threadIdx.x & 63] +
                        red[0][3][threadIdx.x & 63]) / (float)P;
    float q = 0.f;
    for (int p = ph; p < P; p += 4) {
        const float dlt = x[base + (long)p * C] - mean;
        q += dlt * dlt;
    }
    red[1][ph][threadIdx.x & 63] = q;
    __syncthreads();
    const float var = (red[1][0][threadIdx.x & 63] + red[1][1][threadIdx.x & 63] + red[1][2][threadIdx.x & 63] +
                       red[1][3][threadIdx.x & 63]) / (float)P;
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int p = ph; p < P; p += 4) {
        const long i = base + (long)p * C;
        out[i] = gamma[i] * ((x[i] - mean) * rstd) + beta[i];
    }
}

__global__ void add_scaled_rowvec_kernel(float* __restrict__ x, const float* __restrict__ s, const float* __restrict__ vec,
                                         long per_b, int C, long total) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= total) return;
    const long b = i / per_b;
    const int c = (int)(i % C);
    f32x4 v = *(f32x4*)(x + i);
    v += s[b] * *(const f32x4*)(vec + c);
    *(f32x4*)(x + i) = v;
}

__global__ void row_affine_kernel(float* __restrict__ x, const float* __restrict__ rs, const float* __restrict__ table,
                                  long rows, int C, int div, int mod) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= rows * C) return;
    const long r = i / C;
    const int c = (int)(i - r * C);
    f32x4 v = *(f32x4*)(x + i);
    if (rs) v *= rs[r];
    if (table) v += *(const f32x4*)(table + ((r / div) % mod) * (long)C + c);
    *(f32x4*)(x + i) = v;
}

template <typename OT>
int ln_launch(const float* x, const float* g, const float* b, void* y, int64_t rows, int C, float eps, hipStream_t s) {
    const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
    if (C <= 256) hipLaunchKernelGGL((layernorm_kernel<OT, 1>), grid, blk, 0, s, x, g, b, (OT*)y, (long)rows, C, eps);
    else if (C <= 512) hipLaunchKernelGGL((layernorm_kernel<OT, 2>), grid, blk, 0, s, x, g, b, (OT*)y, (long)rows, C, eps);
    else if (C <= 1024) hipLaunchKernelGGL((layernorm_kernel<OT, 4>), grid, blk, 0, s, x, g, b, (OT*)y, (long)rows, C, eps);
    else hipLaunchKernelGGL((layernorm_kernel<OT, 8>), grid, blk, 0, s, x, g, b, (OT*)y, (long)rows, C, eps);
    MAGE_CHECK_LAUNCH("mage_layernorm");
    return MAGE_OK;
}

template <typename T>
int attn_launch(const mage_attn_desc* d, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        // the axial attentions: short sequences on the matrix cores (16-byte aligned 64-byte head segments)
        if (d->nq <= 32 && d->nk <= 32 && d->n_head <= 32 && !mage_options().attn_no_mfma &&
            ((((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v | (uintptr_t)d->out) & 15) == 0)) {
            const int nkb = d->nk <= 16 ? 1 : 2;
            if (d->nq <= 2 && d->n_seq >= 1024 && !mage_options().attn_no_fewq) {      // the incremental step's temporal attention
                const dim3 grid((unsigned)((d->n_seq + 3) / 4));
                if (nkb == 1) hipLaunchKernelGGL((attention_mfma_fewq_kernel<1, T>), grid, dim3(256), 0, s, *d);
                else hipLaunchKernelGGL((attention_mfma_fewq_kernel<2, T>), grid, dim3(256), 0, s, *d);
                MAGE_CHECK_LAUNCH("mage_attention");
                return MAGE_OK;
            }
            const size_t lds = (size_t)16 * nkb * (d->n_head * 32 + 16) * 2;
            const int hpw = (d->n_head + 3) / 4;
#define ATTN_MFMA(NKB, MH) hipLaunchKernelGGL((attention_mfma_kernel<NKB, MH, T>), dim3(d->n_seq), dim3(256), lds, s, *d)
            if (nkb == 1) { if (hpw <= 2) ATTN_MFMA(1, 2); else if (hpw <= 4) ATTN_MFMA(1, 4); else ATTN_MFMA(1, 8); }
            else { if (hpw <= 2) ATTN_MFMA(2, 2); else if (hpw <= 4) ATTN_MFMA(2, 4); else ATTN_MFMA(2, 8); }
#undef ATTN_MFMA
            MAGE_CHECK_LAUNCH("mage_attention");
            return MAGE_OK;
        }
    }
    // heads per workgroup: all of them if K,V fit 64 KiB of LDS, else the largest power of two that does
    int hg = d->n_head;
    while ((long)d->nk * hg * 32 * 2 * sizeof(T) > 64 * 1024 && hg > 1 && hg % 2 == 0) hg /= 2;
    const size_t lds = (size_t)d->nk * hg * 32 * 2 * sizeof(T);
    MAGE_CHECK_ARG(lds <= 64 * 1024 && d->n_head % hg == 0, "mage_attention: nk=%d too large for LDS staging", d->nk);
    const dim3 grid(d->n_seq, d->n_head / hg), blk(256);
    if constexpr (sizeof(T) == 4) {
        if (d->out_split == MAGE_BF16X3 || d->out_split == MAGE_F16X3) {
#define ATTN_SPLIT(NK) \
    do { \
        if (d->out_split == MAGE_BF16X3) hipLaunchKernelGGL((attention_kernel<float, NK, split_bf16>), grid, blk, lds, s, *d, hg); \
        else hipLaunchKernelGGL((attention_kernel<float, NK, split_f16>), grid, blk, lds, s, *d, hg); \
    } while (0)
            if (d->nk <= 16) ATTN_SPLIT(16); else if (d->nk <= 32) ATTN_SPLIT(32); else ATTN_SPLIT(64);
#undef ATTN_SPLIT
            MAGE_CHECK_LAUNCH("mage_attention");
            return MAGE_OK;
        }
    }
    if (d->nk <= 16) hipLaunchKernelGGL((attention_kernel<T, 16>), grid, blk, lds, s, *d, hg);
    else if (d->nk <= 32) hipLaunchKernelGGL((attention_kernel<T, 32>), grid, blk, lds, s, *d, hg);
    else hipLaunchKernelGGL((attention_kernel<T, 64>), grid, blk, lds, s, *d, hg);
    MAGE_CHECK_LAUNCH("mage_attention");
    return MAGE_OK;
}

}  // namespace

extern "C" int mage_layernorm(const float* x, const float* gamma, const float* beta, void* y, int32_t y_dtype,
                              int64_t rows, int32_t C, float eps, void* stream) {
    MAGE_CHECK_ARG(x && gamma && beta && y, "mage_layernorm: null pointer");
    MAGE_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0 && C <= 2048, "mage_layernorm: rows=%ld C=%d unsupported", (long)rows, C);
    if (y_dtype == MAGE_F32) return ln_launch<float>(x, gamma, beta, y, rows, C, eps, (hipStream_t)stream);
    if (y_dtype == MAGE_BF16) return ln_launch<unsigned short>(x, gamma, beta, y, rows, C, eps, (hipStream_t)stream);
    if (y_dtype == MAGE_F16) return ln_launch<f16_t>(x, gamma, beta, y, rows, C, eps, (hipStream_t)stream);
    if (y_dtype == MAGE_BF16X3 || y_dtype == MAGE_F16X3) {
        MAGE_CHECK_ARG(C % 64 == 0 && (((uintptr_t)y) & 255) == 0, "mage_layernorm: split output needs C %% 64 == 0 and y 256-byte aligned");
        if (y_dtype == MAGE_BF16X3) return ln_launch<split_bf16>(x, gamma, beta, y, rows, C, eps, (hipStream_t)stream);
        return ln_launch<split_f16>(x, gamma, beta, y, rows, C, eps, (hipStream_t)stream);
    }
    mage_set_error("mage_layernorm: bad y_dtype %d", y_dtype);
    return MAGE_EINVAL;
}

// ------------------------------------------------------------------------------------ fp32 rows -> split-precision rows
namespace {
template <typename OT>
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, long ldx, OT* __restrict__ y, long ldy, long rows, int C) {
    const int vpr = C >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * vpr) return;
    const long r = i / vpr;
    const int c = (int)(i - r * vpr) << 2;
    store4(y + r * ldy + c, *(const f32x4*)(x + r * ldx + c));
}
}  // namespace

namespace {
template <typename OT>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, long ldx, OT* __restrict__ y, long rows, int C, int relu,
                                                         long group, long group_stride, long off, long inner, long inner_stride,
                                                         float* __restrict__ x_relu) {
    const int vpr = C >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * vpr) return;
    const long r = i / vpr;
    const int c = (int)(i - r * vpr) << 2;
    f32x4 v = *(const f32x4*)(x + r * ldx + c);
    if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (x_relu) *(f32x4*)(x_relu + r * ldx + c) = v;
    const long ig = r % group;
    const long orow = (r / group) * group_stride + (ig / inner) * inner_stride + (ig % inner) + off;
    store4(y + orow * C + c, v);
}
}  // namespace

extern "C" int mage_split_rows(const float* x, int64_t ldx, void* y, int64_t rows, int32_t C, int32_t kind, int32_t relu, int64_t group,
                               int64_t group_stride, int64_t off, int64_t inner, int64_t inner_stride, float* x_relu, void* stream) {
    MAGE_CHECK_ARG(x && y && rows > 0 && C > 0 && C % 64 == 0 && ldx % 4 == 0 && group > 0, "mage_split_rows: C=%d must be a multiple of 64", C);
    MAGE_CHECK_ARG(((((uintptr_t)y) & 255) | (((uintptr_t)x) & 15)) == 0, "mage_split_rows: y must be 256-byte aligned, x 16-byte aligned");
    MAGE_CHECK_ARG(kind == MAGE_BF16X3 || kind == MAGE_F16X3, "mage_split_rows: kind %d is not a split kind", kind);
    if (inner <= 0) {
        inner = group;
        inner_stride = group;
    }
    const dim3 grid((unsigned)((rows * (C >> 2) + 255) / 256));
    if (kind == MAGE_BF16X3)
        hipLaunchKernelGGL((split_rows_kernel<split_bf16>), grid, dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (split_bf16*)y, (long)rows, C, relu,
                           (long)group, (long)group_stride, (long)off, (long)inner, (long)inner_stride, x_relu);
    else
        hipLaunchKernelGGL((split_rows_kernel<split_f16>), grid, dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (split_f16*)y, (long)rows, C, relu,
                           (long)group, (long)group_stride, (long)off, (long)inner, (long)inner_stride, x_relu);
    MAGE_CHECK_LAUNCH("mage_split_rows");
    return MAGE_OK;
}

extern "C" int mage_split(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t C, int32_t kind, void* stream) {
    MAGE_CHECK_ARG(x && y && rows > 0 && C > 0 && C % 64 == 0 && ldx % 4 == 0 && ldy % 128 == 0 && ldy >= 2 * (int64_t)C,
                   "mage_split: C=%d must be a multiple of 64, ldy a multiple of 128 16-bit elements (>= 2C)", C);
    MAGE_CHECK_ARG(((((uintptr_t)y) & 255) | (((uintptr_t)x) & 15)) == 0, "mage_split: y must be 256-byte aligned, x 16-byte aligned");
    MAGE_CHECK_ARG(kind == MAGE_BF16X3 || kind == MAGE_F16X3, "mage_split: kind %d is not a split kind", kind);
    const dim3 grid((unsigned)((rows * (C >> 2) + 255) / 256));
    if (kind == MAGE_BF16X3)
        hipLaunchKernelGGL((split_kernel<split_bf16>), grid, dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (split_bf16*)y, (long)(ldy >> 1), (long)rows, C);
    else
        hipLaunchKernelGGL((split_kernel<split_f16>), grid, dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (split_f16*)y, (long)(ldy >> 1), (long)rows, C);
    MAGE_CHECK_LAUNCH("mage_split");
    return MAGE_OK;
}

namespace {
int attn_split_launch(const mage_attn_desc* d, hipStream_t s) {
    MAGE_CHECK_ARG(d->nq <= 32 && d->nk <= 32 && d->n_head <= 32 && d->n_head % 2 == 0 && d->out_split == MAGE_F16X3 && d->drop_p == 0.f,
                   "mage_attention: split (f16x3) q/k/v: nq, nk <= 32, an even head count, split output");
    MAGE_CHECK_ARG((d->ldq | d->ldk | d->ldv | d->ldo) % 128 == 0 && ((((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v | (uintptr_t)d->out) & 255) == 0),
                   "mage_attention: split operands need leading dimensions that are multiples of 128 16-bit elements and 256-byte aligned bases");
    const int nkb = d->nk <= 16 ? 1 : 2;
    if (d->nq <= 2 && d->n_seq >= 1024 && !mage_options().attn_no_fewq) {      // the incremental step's temporal attention
        const dim3 grid((unsigned)((d->n_seq + 3) / 4));
        if (nkb == 1) hipLaunchKernelGGL((attention_mfma_split_fewq_kernel<1>), grid, dim3(256), 0, s, *d);
        else hipLaunchKernelGGL((attention_mfma_split_fewq_kernel<2>), grid, dim3(256), 0, s, *d);
        MAGE_CHECK_LAUNCH("mage_attention");
        return MAGE_OK;
    }
    const size_t lds = (size_t)16 * nkb * (d->n_head * 64 + 16) * 2;
    const int hpw = (d->n_head + 3) / 4;
#define ATTN_S(NKB, MH) \
    do { \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attention_mfma_split_kernel<NKB, MH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((attention_mfma_split_kernel<NKB, MH>), dim3(d->n_seq), dim3(256), lds, s, *d); \
    } while (0)
    if (nkb == 1) { if (hpw <= 2) ATTN_S(1, 2); else if (hpw <= 4) ATTN_S(1, 4); else ATTN_S(1, 8); }
    else { if (hpw <= 2) ATTN_S(2, 2); else if (hpw <= 4) ATTN_S(2, 4); else ATTN_S(2, 8); }
#undef ATTN_S
    MAGE_CHECK_LAUNCH("mage_attention");
    return MAGE_OK;
}
}  // namespace

extern "C" int mage_attention(const mage_attn_desc* d, void* stream) {
    MAGE_CHECK_ARG(d && d->q && d->k && d->v && d->out, "mage_attention: null pointer");
    MAGE_CHECK_ARG(d->nk >= 1 && d->nk <= 64, "mage_attention: nk=%d outside [1, 64]", d->nk);
    MAGE_CHECK_ARG(d->nq >= 1 && d->n_seq >= 1 && d->n_head >= 1 && d->inner >= 1, "mage_attention: bad sizes");
    MAGE_CHECK_ARG((d->ldq | d->ldk | d->ldv | d->ldo) % 8 == 0, "mage_attention: leading dims must be multiples of 8");
    MAGE_CHECK_ARG(!d->kv_len || d->kv_len_div >= 1, "mage_attention: kv_len_div must be >= 1");
    MAGE_CHECK_ARG((d->o_axis_stride == 0 && d->o_outer_stride == 0) || d->o_axis_stride > 0, "mage_attention: o_axis_stride must be > 0 when an output row map is given");
    MAGE_CHECK_ARG(d->drop_p >= 0.f && d->drop_p < 1.f && (d->drop_p == 0.f || (d->dtype == MAGE_F32 && d->out_split == 0)),
                   "mage_attention: drop_p=%g needs fp32 q/k/v (the thread-per-query kernel) and 0 <= p < 1", (double)d->drop_p);
    MAGE_CHECK_ARG(d->out_split == 0 || ((d->out_split == MAGE_BF16X3 || d->out_split == MAGE_F16X3) && (d->dtype == MAGE_F32 || d->dtype == MAGE_F16X3) &&
                                         (d->n_head * 32) % 64 == 0 && d->ldo % 128 == 0 && (((uintptr_t)d->out) & 255) == 0),
                   "mage_attention: out_split needs fp32 q/k/v, an even head count, ldo a multiple of 128 16-bit elements, out 256-byte aligned");
    if (d->dtype == MAGE_F32) return attn_launch<float>(d, (hipStream_t)stream);
    if (d->dtype == MAGE_BF16) return attn_launch<unsigned short>(d, (hipStream_t)stream);
    if (d->dtype == MAGE_F16) return attn_launch<f16_t>(d, (hipStream_t)stream);
    if (d->dtype == MAGE_F16X3) return attn_split_launch(d, (hipStream_t)stream);
    mage_set_error("mage_attention: bad dtype %d", d->dtype);
    return MAGE_EINVAL;
}

extern "C" int mage_adain(const float* x, const float* gamma, const float* beta, float* out, int32_t B, int32_t P,
                          int32_t C, float eps, void* stream) {
    MAGE_CHECK_ARG(x && gamma && beta && out, "mage_adain: null pointer");
    MAGE_CHECK_ARG(B > 0 && P > 0 && C > 0 && C % 64 == 0, "mage_adain: C=%d must be a multiple of 64", C);
    hipLaunchKernelGGL(adain_kernel, dim3(B, C / 64), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, out, P, C, eps);
    MAGE_CHECK_LAUNCH("mage_adain");
    return MAGE_OK;
}

extern "C" int mage_add_scaled_rowvec(float* x, const float* s, const float* vec, int32_t B, int32_t P, int32_t C,
                                      void* stream) {
    MAGE_CHECK_ARG(x && s && vec, "mage_add_scaled_rowvec: null pointer");
    MAGE_CHECK_ARG(B > 0 && P > 0 && C > 0 && C % 4 == 0, "mage_add_scaled_rowvec: C=%d must be a multiple of 4", C);
    const long total = (long)B * P * C;
    hipLaunchKernelGGL(add_scaled_rowvec_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, s, vec, (long)P * C, C, total);
    MAGE_CHECK_LAUNCH("mage_add_scaled_rowvec");
    return MAGE_OK;
}

// Caption bookkeeping of the text encoder in one launch (mage_model.py:233-239: `text != padding_idx`, its row sums): keep[b*S + s] = 1.0 where
// ids[b][s] != padding_idx else 0.0 (the row scale that zeroes padded rows), kv_len[b] = number of kept tokens (the key-padding length).
namespace {
__global__ __launch_bounds__(64) void caption_mask_kernel(const int64_t* __restrict__ ids, int S, long padding_idx, int* __restrict__ kv_len,
                                                          float* __restrict__ keep) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int cnt = 0;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const bool k = s < S && ids[(long)b * S + s] != padding_idx;
        if (s < S && keep) keep[(long)b * S + s] = k ? 1.0f : 0.0f;
        cnt += __popcll(__ballot(k));
    }
    if (lane == 0 && kv_len) kv_len[b] = cnt;
}
}  // namespace

extern "C" int mage_caption_mask(const int64_t* ids, int32_t B, int32_t S, int64_t padding_idx, int32_t* kv_len, float* keep, void* stream) {
    MAGE_CHECK_ARG(ids && B > 0 && S > 0 && (kv_len || keep), "mage_caption_mask: bad arguments");
    hipLaunchKernelGGL(caption_mask_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, ids, S, (long)padding_idx, kv_len, keep);
    MAGE_CHECK_LAUNCH("mage_caption_mask");
    return MAGE_OK;
}

// Strided block copy on the stream (hipMemcpy2DAsync, device to device): `height` rows of `width_bytes`, row pitches in bytes.  Output
// assembly without a compute kernel: the passed-through first frame and the decoded frames into the [B, L, C, H, W] result (mage_model.py:691).
extern "C" int mage_copy2d(void* dst, int64_t dst_pitch, const void* src, int64_t src_pitch, int64_t width_bytes, int64_t height, void* stream) {
    MAGE_CHECK_ARG(dst && src && width_bytes > 0 && height > 0 && dst_pitch >= width_bytes && src_pitch >= width_bytes, "mage_copy2d: bad arguments");
    const hipError_t e = hipMemcpy2DAsync(dst, (size_t)dst_pitch, src, (size_t)src_pitch, (size_t)width_bytes, (size_t)height, hipMemcpyDeviceToDevice,
                                          (hipStream_t)stream);
    if (e != hipSuccess) {
        mage_set_error("mage_copy2d: %s", hipGetErrorString(e));
        return MAGE_EHIP;
    }
    return MAGE_OK;
}

extern "C" int mage_row_affine(float* x, const float* rs, const float* table, int64_t rows, int32_t C, int32_t div, int32_t mod,
                               void* stream) {
    MAGE_CHECK_ARG(x && rows > 0 && C > 0 && C % 4 == 0, "mage_row_affine: bad arguments");
    MAGE_CHECK_ARG(!table || (div >= 1 && mod >= 1), "mage_row_affine: bad div/mod");
    const long total = (long)rows * C;
    hipLaunchKernelGGL(row_affine_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, rs,
                       table, (long)rows, C, div, mod);
    MAGE_CHECK_LAUNCH("mage_row_affine");
    return MAGE_OK;
}

// ------------------------------------------------------------------------------------ GroupNorm + SiLU (MAGE+ head)
namespace {

// stats[b][g] = {mean, rstd} over rows_per_sample rows x (C/groups) channels; one workgroup per (b, g), two passes,
// fixed-order reductions (deterministic).
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, long sample_stride_rows, long row_off,
                                                       int rows_per_sample, int C, int groups, float eps,
                                                       float* __restrict__ stats) {
    __shared__ double red[4];
    const int b = blockIdx.x, g = blockIdx.y, cpg = C / groups;
    const float* base = x + ((long)b * sample_stride_rows + row_off) * C + g * cpg;
    const long n = (long)rows_per_sample * cpg;
    auto block_sum = [&](double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    double s = 0.0;
    for (int r = threadIdx.x; r < rows_per_sample; r += 256)
        for (int c = 0; c < cpg; ++c) s += (double)base[(long)r * C + c];
    const double mean = block_sum(s) / (double)n;
    double q = 0.0;
    for (int r = threadIdx.x; r < rows_per_sample; r += 256)
        for (int c = 0; c < cpg; ++c) {
            const double dlt = (double)base[(long)r * C + c] - mean;
            q += dlt * dlt;
        }
    const double var = block_sum(q) / (double)n;
    if (threadIdx.x == 0) {
        stats[((long)b * groups + g) * 2] = (float)mean;
        stats[((long)b * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// act: 0 none, 1 ReLU, 2 SiLU.  residual (optional, fp32, packed like the logical output) is added after the affine, before act.
// Output row of (sample b, row r): b*y_sample_stride_rows + y_row_off + r.
template <typename OT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, long sample_stride_rows, long row_off,
                                                       int rows_per_sample, int C, int groups, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ residual, int act, OT* __restrict__ y,
                                                       long y_sample_stride_rows, long y_row_off, long total) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    const long orow = i / C;
    const int c = (int)(i - orow * C);
    const long b = orow / rows_per_sample, r = orow - b * rows_per_sample;
    const int cpg = C / groups;
    const f32x4 v = *(const f32x4*)(x + ((b * sample_stride_rows + row_off + r) * C + c));
    const f32x4 gm = *(const f32x4*)(gamma + c), bt = *(const f32x4*)(beta + c);
    f32x4 res = f32x4{0.f, 0.f, 0.f, 0.f};
    if (residual) res = *(const f32x4*)(residual + i);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int g = (c + e) / cpg;                                  // a vector may straddle groups when C/groups < 4
        const float mean = stats[(b * groups + g) * 2], rstd = stats[(b * groups + g) * 2 + 1];
        const float t = (v[e] - mean) * rstd * gm[e] + bt[e] + res[e];
        o[e] = act == 2 ? t / (1.0f + expf(-t)) : act == 1 ? fmaxf(t, 0.f) : t;
    }
    store4(y + ((b * y_sample_stride_rows + y_row_off + r) * C + c), o);
}

int gn_launch(const char* who, const float* x, int64_t sample_stride_rows, int64_t row_off, int32_t n_samples, int32_t rows_per_sample,
              int32_t C, int32_t groups, const float* gamma, const float* beta, float eps, float* stats, const float* residual,
              int32_t act, void* y, int32_t y_dtype, int64_t y_sample_stride_rows, int64_t y_row_off, hipStream_t s) {
    hipLaunchKernelGGL(gn_stats_kernel, dim3(n_samples, groups), dim3(256), 0, s, x, (long)sample_stride_rows, (long)row_off,
                       rows_per_sample, C, groups, eps, stats);
    const long total = (long)n_samples * rows_per_sample * C;
    const dim3 grid((unsigned)((total / 4 + 255) / 256));
    if (y_dtype == MAGE_F32)
        hipLaunchKernelGGL((gn_apply_kernel<float>), grid, dim3(256), 0, s, x, (long)sample_stride_rows, (long)row_off, rows_per_sample,
                           C, groups, stats, gamma, beta, residual, act, (float*)y, (long)y_sample_stride_rows, (long)y_row_off, total);
    else if (y_dtype == MAGE_BF16)
        hipLaunchKernelGGL((gn_apply_kernel<unsigned short>), grid, dim3(256), 0, s, x, (long)sample_stride_rows, (long)row_off,
                           rows_per_sample, C, groups, stats, gamma, beta, residual, act, (unsigned short*)y,
                           (long)y_sample_stride_rows, (long)y_row_off, total);
    else if (y_dtype == MAGE_F16)                      // the f16 mode on the latent (MAGE+) path: the GroupNorm + SiLU head's rows (round 6)
        hipLaunchKernelGGL((gn_apply_kernel<f16_t>), grid, dim3(256), 0, s, x, (long)sample_stride_rows, (long)row_off,
                           rows_per_sample, C, groups, stats, gamma, beta, residual, act, (f16_t*)y,
                           (long)y_sample_stride_rows, (long)y_row_off, total);
    else {
        mage_set_error("%s: bad y_dtype %d", who, y_dtype);
        return MAGE_EINVAL;
    }
    MAGE_CHECK_LAUNCH(who);
    return MAGE_OK;
}

}  // namespace

extern "C" int mage_groupnorm_silu(const float* x, int64_t sample_stride_rows, int64_t row_off, int32_t n_samples,
                                   int32_t rows_per_sample, int32_t C, int32_t groups, const float* gamma, const float* beta,
                                   float eps, float* stats, void* y, int32_t y_dtype, void* stream) {
    MAGE_CHECK_ARG(x && gamma && beta && stats && y, "mage_groupnorm_silu: null pointer");
    MAGE_CHECK_ARG(n_samples > 0 && rows_per_sample > 0 && groups > 0 && C % groups == 0 && C % 4 == 0,
                   "mage_groupnorm_silu: C=%d groups=%d unsupported", C, groups);
    return gn_launch("mage_groupnorm_silu", x, sample_stride_rows, row_off, n_samples, rows_per_sample, C, groups, gamma, beta, eps,
                     stats, nullptr, 2, y, y_dtype, rows_per_sample, 0, (hipStream_t)stream);
}

extern "C" int mage_groupnorm_act(const float* x, int64_t sample_stride_rows, int64_t row_off, int32_t n_samples,
                                  int32_t rows_per_sample, int32_t C, int32_t groups, const float* gamma, const float* beta,
                                  float eps, float* stats, const float* residual, int32_t act, void* y, int32_t y_dtype,
                                  int64_t y_sample_stride_rows, int64_t y_row_off, void* stream) {
    MAGE_CHECK_ARG(x && gamma && beta && stats && y, "mage_groupnorm_act: null pointer");
    MAGE_CHECK_ARG(n_samples > 0 && rows_per_sample > 0 && groups > 0 && C % groups == 0 && C % 4 == 0,
                   "mage_groupnorm_act: C=%d groups=%d unsupported", C, groups);
    MAGE_CHECK_ARG(act >= 0 && act <= 2, "mage_groupnorm_act: act=%d (0 none, 1 relu, 2 silu)", act);
    return gn_launch("mage_groupnorm_act", x, sample_stride_rows, row_off, n_samples, rows_per_sample, C, groups, gamma, beta, eps,
                     stats, residual, act, y, y_dtype, y_sample_stride_rows, y_row_off, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------ reparameterisation + KL summand
namespace {
__global__ __launch_bounds__(256) void reparam_kl_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                         const float* __restrict__ eps, float* __restrict__ out,
                                                         float* __restrict__ kl_sum, long n) {
    __shared__ double red[4];
    const long base = (long)blockIdx.x * n;
    double acc = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float m = mu[base + i], lv = logvar[base + i];
        out[base + i] = eps[base + i] * expf(0.5f * lv) + m;
        acc += (double)(1.0f + lv - m * m - expf(lv));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) kl_sum[blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}
}  // namespace

extern "C" int mage_reparam_kl(const float* mu, const float* logvar, const float* eps, float* out, float* kl_sum, int32_t B,
                               int64_t n, void* stream) {
    MAGE_CHECK_ARG(mu && logvar && eps && out && kl_sum && B > 0 && n > 0, "mage_reparam_kl: bad arguments");
    hipLaunchKernelGGL(reparam_kl_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps, out, kl_sum, (long)n);
    MAGE_CHECK_LAUNCH("mage_reparam_kl");
    return MAGE_OK;
}

// ------------------------------------------------------------------------------------ mean squared error (MAGE+ latent loss)
namespace {
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                          long rows, int cols, double* __restrict__ partial) {
    __shared__ double red[4];
    double acc = 0.0;
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        const float dlt = a[r * lda + c] - b[r * ldb + c];
        acc += (double)(dlt * dlt);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void mse_final_kernel(const double* __restrict__ partial, int n, double inv_count, float* __restrict__ out) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += partial[i];          // fixed order: deterministic
    out[0] = (float)(s * inv_count);
}
}  // namespace

extern "C" int mage_mse(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int32_t cols, double* workspace,
                        float* out, void* stream) {
    MAGE_CHECK_ARG(a && b && workspace && out && rows > 0 && cols > 0, "mage_mse: bad arguments");
    const int nblk = (int)((rows * cols + 255) / 256 < 256 ? (rows * cols + 255) / 256 : 256);
    hipLaunchKernelGGL(mse_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a, (long)lda, b, (long)ldb, (long)rows, cols,
                       workspace);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, workspace, nblk, 1.0 / ((double)rows * cols), out);
    MAGE_CHECK_LAUNCH("mage_mse");
    return MAGE_OK;
}

// ------------------------------------------------------------------------------------ LayerNorm statistics from GEMM partial sums
namespace {
__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ part, long rows, int n_slices, float inv_c, float eps,
                                                       float* __restrict__ stats) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float mean, rstd;
    mage_ln_stats_row((const float2*)part + r, rows, n_slices, inv_c, eps, mean, rstd);          // slice-major: each slice's loads are coalesced
    *(float2*)(stats + 2 * r) = float2{mean, rstd};
}
}  // namespace

// (mean, rstd) of bf16 rows, one wave per row: the first LayerNorm of a decoder pass when the residual stream starts in bf16 (the frame
// fill and context_linear write bf16 rows; the Linear that follows takes (rows, stats) like every later one).  Sums in fp32 in a fixed
// order (lane: its 8-column groups left to right; then the xor tree), the closing formula is mage_ln_stats'.
namespace {
template <typename HT>                                  // unsigned short = bf16 rows, f16_t = f16 rows
__global__ __launch_bounds__(256) void row_stats_kernel(const unsigned short* __restrict__ x, long rows, int C, long ldx, float inv_c, float eps,
                                                        float* __restrict__ stats) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const unsigned short* p = x + r * ldx;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
        const uint4 v = *(const uint4*)(p + c);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 ab = unpack16x2<HT>(w[j]);
            const float a = ab.x, b = ab.y;
            s1 = __fadd_rn(s1, __fadd_rn(a, b));
            s2 = __fadd_rn(s2, __fmaf_rn(a, a, __fmul_rn(b, b)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 = __fadd_rn(s1, __shfl_xor(s1, o));
        s2 = __fadd_rn(s2, __shfl_xor(s2, o));
    }
    if (lane == 0) {
        const float mean = __fmul_rn(s1, inv_c);
        const float var = fmaxf(__fmaf_rn(-mean, mean, __fmul_rn(s2, inv_c)), 0.f);
        *(float2*)(stats + 2 * r) = float2{mean, __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(var, eps)))};
    }
}
}  // namespace

extern "C" int mage_row_stats(const void* x, int32_t dtype, int64_t rows, int32_t C, int64_t ldx, float eps, float* stats, void* stream) {
    MAGE_CHECK_ARG(x && stats && rows > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldx % 8 == 0 && (((uintptr_t)x) & 15) == 0 &&
                       (dtype == MAGE_BF16 || dtype == MAGE_F16),
                   "mage_row_stats: bf16 or f16 rows, C and ldx multiples of 8, x 16-byte aligned");
    if (dtype == MAGE_F16)
        hipLaunchKernelGGL(row_stats_kernel<f16_t>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x,
                           (long)rows, C, (long)ldx, 1.0f / (float)C, eps, stats);
    else
        hipLaunchKernelGGL(row_stats_kernel<unsigned short>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)x, (long)rows, C, (long)ldx, 1.0f / (float)C, eps, stats);
    MAGE_CHECK_LAUNCH("mage_row_stats");
    return MAGE_OK;
}

extern "C" int mage_ln_stats(const float* part, int64_t rows, int32_t n_slices, int32_t C, float eps, float* stats, void* stream) {
    MAGE_CHECK_ARG(part && stats && rows > 0 && n_slices > 0 && C > 0, "mage_ln_stats: bad arguments");
    hipLaunchKernelGGL(ln_stats_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, (long)rows, n_slices,
                       1.0f / (float)C, eps, stats);
    MAGE_CHECK_LAUNCH("mage_ln_stats");
    return MAGE_OK;
}
