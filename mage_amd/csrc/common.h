// Shared helpers for the libmage_hip.so kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mage_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define MAGE_MAX_DEVICES 16
void mage_set_error(const char* fmt, ...);
int mage_device_index();               // current HIP device, or -1 (per-device caches of launch attributes are indexed by it)
const void* mage_zero_page();          // device pointer, 16384 zero bytes (padding source of the gather loads; a zero bias vector of up to 4096
                                       // columns); null before mage_init
int* mage_error_word();                // device pointer to the deferred-error word of the current device (mage_check_device_errors)
enum { MAGE_DEVERR_EMBEDDING_ID = 1, MAGE_DEVERR_CE_TARGET = 2 };
// Kernel-selection switches (tuning, A/B tests, bisecting): ONE table, filled once from the MAGE_* environment variables of the same names
// (upper-cased, "MAGE_" prefix) the first time it is asked for, changed at run time only through mage_set_option (include/mage_hip.h lists
// them).  No dispatch function reads the environment.
struct MageOptions {
    int gemm_no_4w;              // 1: the one-wave-per-SIMD GEMM kernels (gemm4.hip) are not used
    int gemm_no_4h;              // 1: the split-half form of that kernel (gemm4h.hip: epilogue under the K loop) is not used (gemm4_kernel runs)
    int gemm_4h_plain;           // 1: gemm4h_kernel also takes the forms without QuickGELU (QKV); default: the QuickGELU forms (c_fc) only
    int gemm4_train_forms;       // 1: training's two c_fc forms on the one-wave-per-SIMD kernel (default: the 8-phase kernel, faster there)
    int gemm_no_8phase;          // 1: the 8-phase ping-pong kernel is not used (lockstep kernel instead)
    int gemm_no_taps8;           // 1: padded-taps convolutions take the generic gather kernel
    int gemm_no_narrow;          // 1: no 256 x 64 narrow tiles
    int gemm_no_narrow_few;      // 1: few-rows x + Linear(.) stays on the 128 x 256 tile
    int gemm_no_small;           // 1: the few-rows kernel (gemm_small_kernel) is not used
    int gemm_small_m;            // rows up to which the few-rows kernel is considered (default 1024)
    int gemm_stagger_groups, gemm_stagger_percent, gemm_stagger_forced;      // staggered start of the 8-wave kernels (MAGE_GEMM_STAGGER="G,percent")
    int gemm4_stagger_groups, gemm4_stagger_percent;                         // ... of the one-wave-per-SIMD kernel (MAGE_GEMM4_STAGGER)
    int attn_no_mfma;            // 1: attention on the thread-per-query kernels only
    int attn_no_fewq;            // 1: the few-query (incremental step) attention kernels are not used
    int vq_no_mfma;              // 1: the quantiser's fp64-MFMA kernel is not used
    int conv_no_tile;            // 1: the 64-channel 3x3 tile convolution (conv_tile.hip) is not used (the implicit-GEMM kernels run)
};
const MageOptions& mage_options();
// raise a deferred error from a kernel: the first one wins, the offending value and the bound are kept for the message
__device__ __forceinline__ void mage_raise(int* word, int code, long value, int bound) {
    if (atomicCAS(word, 0, code) == 0) {
        word[1] = (int)value;
        word[2] = bound;
    }
}

// Debug build (`make debug`: -DMAGE_DEBUG=1, host code under AddressSanitizer): device-side invariants -- LDS offsets inside their rings and
// windows, tile indices inside the tile list, table rows inside their tables -- as device asserts (a failed one aborts the kernel and the
// next synchronisation reports it with file and line).  Compiled out of the release library.
#if defined(MAGE_DEBUG) && MAGE_DEBUG
#include <cassert>
#define MAGE_DASSERT(cond) assert(cond)
#else
#define MAGE_DASSERT(cond) ((void)0)
#endif

#define MAGE_CHECK_ARG(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) {                                              \
            mage_set_error(__VA_ARGS__);                            \
            return MAGE_EINVAL;                                     \
        }                                                           \
    } while (0)

#define MAGE_CHECK_LAUNCH(name)                                                         \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            mage_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
            return MAGE_EHIP;                                                           \
        }                                                                               \
    } while (0)

// round-to-nearest-even fp32 -> bf16 (same rounding as torch's .to(torch.bfloat16))
__device__ __forceinline__ unsigned short f2bf_bits(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf_bits2f(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<unsigned short>(unsigned short v) { return bf_bits2f(v); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ unsigned short from_f32<unsigned short>(float v) { return f2bf_bits(v); }

// MAGE_F16 storage: IEEE half as `_Float16` (a type of its own, so that every kernel templated on its 16-bit element type picks the f16
// conversions by overload; `unsigned short` stays the bf16 storage type).  fp32 -> f16 is the hardware's round-to-nearest-even
// (v_cvt_pk_f16_f32); |x| > 65504 becomes inf -- the single-pass f16 mode is for tensors of the decoder stack (weights ~ N(0, 0.02..0.06),
// LayerNorm-ed activations, a residual stream of O(10)), not for arbitrary data.
typedef _Float16 f16_t;
template <> __device__ __forceinline__ float to_f32<f16_t>(f16_t v) { return (float)v; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }

// load / store 4 consecutive elements as fp32
__device__ __forceinline__ f32x4 load4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 load4(const unsigned short* p) {
    uint2 r = *(const uint2*)p;
    f32x4 o;
    o[0] = __uint_as_float(r.x << 16);
    o[1] = __uint_as_float(r.x & 0xffff0000u);
    o[2] = __uint_as_float(r.y << 16);
    o[3] = __uint_as_float(r.y & 0xffff0000u);
    return o;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *(f32x4*)p = v; }
// two fp32 -> packed bf16x2 with the hardware converter (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN-safe)
// (mean, rstd) of a row from the producer GEMM's per-slice partial sums (sum, sum of squares): ONE definition for mage_ln_stats and for the
// few-rows GEMM that folds it into its prologue, explicit roundings: the same bits from both.
// (p: the row's first partial sum; consecutive slices are `stride` float2 apart: slice-major ln_part, mage_gemm_desc::ln_part_rows)
__device__ __forceinline__ void mage_ln_stats_row(const float2* __restrict__ p, long stride, int n_slices, float inv_c, float eps, float& mean, float& rstd) {
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < n_slices; ++i) {
        const float2 v = p[(long)i * stride];
        s1 = __fadd_rn(s1, v.x);
        s2 = __fadd_rn(s2, v.y);
    }
    mean = __fmul_rn(s1, inv_c);
    const float var = fmaxf(__fmaf_rn(-mean, mean, __fmul_rn(s2, inv_c)), 0.f);
    rstd = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(var, eps)));
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned int pack_bf16x2(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ void store4(unsigned short* p, f32x4 v) {
    uint2 r;
    r.x = pack_bf16x2(v[0], v[1]);
    r.y = pack_bf16x2(v[2], v[3]);
    *(uint2*)p = r;
}

// 8 consecutive elements (16 bytes of bf16) in one store
__device__ __forceinline__ void store8(unsigned short* p, f32x4 a, f32x4 b) {
    uint4 r;
    r.x = pack_bf16x2(a[0], a[1]);
    r.y = pack_bf16x2(a[2], a[3]);
    r.z = pack_bf16x2(b[0], b[1]);
    r.w = pack_bf16x2(b[2], b[3]);
    *(uint4*)p = r;
}
__device__ __forceinline__ void store8(float* p, f32x4 a, f32x4 b) { *(f32x4*)p = a; *(f32x4*)(p + 4) = b; }

// ---- f16 rows (MAGE_F16): the same helpers on `f16_t`
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ unsigned int pack_f16x2(float a, float b) {
    const f16x2_t v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float2 unpack_f16x2(unsigned int u) {
    const f16x2_t v = __builtin_bit_cast(f16x2_t, u);
    return float2{(float)v[0], (float)v[1]};
}
__device__ __forceinline__ float2 unpack_bf16x2(unsigned int u) { return float2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
// two fp32 -> one packed pair of the 16-bit type T / back
template <typename T> __device__ __forceinline__ unsigned int pack16x2(float a, float b) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, unsigned short)) return pack_f16x2(a, b);
    else return pack_bf16x2(a, b);
}
template <typename T> __device__ __forceinline__ float2 unpack16x2(unsigned int u) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, unsigned short)) return unpack_f16x2(u);
    else return unpack_bf16x2(u);
}
// 4 packed 16-bit values (8 bytes) -> fp32
template <typename T> __device__ __forceinline__ f32x4 widen4(uint2 r) {
    const float2 a = unpack16x2<T>(r.x), b = unpack16x2<T>(r.y);
    return f32x4{a.x, a.y, b.x, b.y};
}
// v_mfma_f32_16x16x32 on 8 packed 16-bit values per operand, by the storage type: bf16 (unsigned short) or f16 (f16_t); same rate
template <typename T> __device__ __forceinline__ f32x4 mfma16x16x32(const u32x4& a, const u32x4& b, const f32x4& c) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, unsigned short))
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 load4(const f16_t* p) { return widen4<f16_t>(*(const uint2*)p); }
__device__ __forceinline__ void store4(f16_t* p, f32x4 v) { *(uint2*)p = uint2{pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3])}; }
__device__ __forceinline__ void store8(f16_t* p, f32x4 a, f32x4 b) {
    *(uint4*)p = uint4{pack_f16x2(a[0], a[1]), pack_f16x2(a[2], a[3]), pack_f16x2(b[0], b[1]), pack_f16x2(b[2], b[3])};
}

// ---- split-precision storage (MAGE_BF16X3 / MAGE_F16X3, include/mage_hip.h) -------------------------------------------------
// A logical fp32 matrix [rows, C] (C % 64 == 0) kept as TWO 16-bit pieces per element, x ~ hi + lo / LO_SCALE, laid out per row as
// 64-column slabs  [hi(64) | lo(64)]  (256 bytes per slab: the GEMM's K slab of the hi piece, then of the lo piece).
//   bf16 pieces: hi = bf16(x), lo = bf16(x - hi)                 (8 + 8 significand bits, fp32 exponent range)
//   f16  pieces: hi = f16(x),  lo = f16((x - hi) * 2^11)         (11 + 11 significand bits; |x| clamped to 65504)
// One logical element occupies 4 bytes, so `T* row = base + r * C` is the row start for T = split_bf16 / split_f16 as it is for
// float, and the slab-relative address of column c follows from the byte address alone when the base is 256-byte aligned:
// that is what store4 / store8 below use, so every kernel templated on its output type writes split rows unchanged.
// max(x, 0) on two packed 16-bit floats (bf16 or f16: a negative value is a negative int16, -0 included)
typedef short mage_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu16x2(unsigned v) {
    const mage_s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(mage_s16x2, v), z));
}

struct split_bf16 { unsigned int pair; };
struct split_f16 { unsigned int pair; };
#define MAGE_F16_LO_SCALE 2048.0f
// KIND 1 = bf16 pieces, 2 = f16 pieces: two values -> packed hi pair, packed lo pair
template <int KIND>
__device__ __forceinline__ void split_pack2(float a, float b, unsigned int& hi, unsigned int& lo) {
    if constexpr (KIND == 1) {
        hi = pack_bf16x2(a, b);
        lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
    } else {
        a = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);
        b = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
        const f16x2_t h = {(_Float16)a, (_Float16)b};
        const f16x2_t l = {(_Float16)((a - (float)h[0]) * MAGE_F16_LO_SCALE), (_Float16)((b - (float)h[1]) * MAGE_F16_LO_SCALE)};
        hi = __builtin_bit_cast(unsigned int, h);
        lo = __builtin_bit_cast(unsigned int, l);
    }
}
template <int KIND>
__device__ __forceinline__ void split_store4(void* p, f32x4 v) {
    const uintptr_t a = (uintptr_t)p;
    char* q = (char*)((a & ~(uintptr_t)255) + ((a & 255) >> 1));
    uint2 h, l;
    split_pack2<KIND>(v[0], v[1], h.x, l.x);
    split_pack2<KIND>(v[2], v[3], h.y, l.y);
    *(uint2*)q = h;
    *(uint2*)(q + 128) = l;
}
__device__ __forceinline__ void store4(split_bf16* p, f32x4 v) { split_store4<1>(p, v); }
__device__ __forceinline__ void store4(split_f16* p, f32x4 v) { split_store4<2>(p, v); }
template <int KIND>
__device__ __forceinline__ void split_store8(void* p, f32x4 v0, f32x4 v1) {
    const uintptr_t a = (uintptr_t)p;
    char* q = (char*)((a & ~(uintptr_t)255) + ((a & 255) >> 1));
    uint4 h, l;
    split_pack2<KIND>(v0[0], v0[1], h.x, l.x);
    split_pack2<KIND>(v0[2], v0[3], h.y, l.y);
    split_pack2<KIND>(v1[0], v1[1], h.z, l.z);
    split_pack2<KIND>(v1[2], v1[3], h.w, l.w);
    *(uint4*)q = h;
    *(uint4*)(q + 128) = l;
}
__device__ __forceinline__ void store8(split_bf16* p, f32x4 a, f32x4 b) { split_store8<1>(p, a, b); }
__device__ __forceinline__ void store8(split_f16* p, f32x4 a, f32x4 b) { split_store8<2>(p, a, b); }

// stateless dropout masks: keep(i) = hash32(seed * 0x9e3779b97f4a7c15 + i) >= p * 2^32 (mage_dropout, mage_attention's drop_p)
__device__ __forceinline__ unsigned hash32(unsigned long long v) {
    v ^= v >> 33;
    v *= 0xff51afd7ed558ccdULL;
    v ^= v >> 33;
    v *= 0xc4ceb9fe1a85ec53ULL;
    v ^= v >> 33;
    return (unsigned)v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
