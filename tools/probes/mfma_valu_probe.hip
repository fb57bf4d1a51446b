// How many vector instructions does ONE wave per SIMD issue for free between back-to-back independent v_mfma_f32_16x16x32_bf16?
// One workgroup of 256 threads per CU (4 waves, one per SIMD, launch_bounds(256) => no second resident), 64 independent MFMAs per loop
// iteration on literal AGPRs, K filler instructions of a kind after every MFMA.  Prints shader cycles per MFMA (s_memtime) and wall time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_valu_probe.hip -o tools/probes/mfma_valu_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <typename F, int... I>
__device__ __forceinline__ void for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) { for_impl(f, std::make_integer_sequence<int, N>{}); }


template <int I, bool Z> __device__ __forceinline__ void mf(const u32x4& w, const u32x4& x) {
    if constexpr (Z) asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(w), "v"(x), "i"(4 * I), "i"(4 * I + 3));
    else asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(x), "i"(4 * I), "i"(4 * I + 3));
}
template <int KIND, int N> __device__ __forceinline__ void filler(float (&r)[8], float seed, unsigned& pk, u32x4& ld, unsigned la) {
    constexpr int e = N & 7;
    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[e]) : "v"(seed));
    if constexpr (KIND == 1) asm volatile("v_mul_f32 %0, 0x3f800001, %0" : "+v"(r[e]));
    if constexpr (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(r[e]));
    if constexpr (KIND == 3) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(r[e]) : "i"(128 + (N & 127)));
    if constexpr (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(r[e]), "v"(r[(e + 1) & 7]));
    if constexpr (KIND == 5) {
        constexpr int s = N & 7;
        if constexpr (s == 0 || s == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[e]) : "v"(seed));
        if constexpr (s == 2 || s == 6) asm volatile("v_mul_f32 %0, 0x3f800001, %0" : "+v"(r[e]));
        if constexpr (s == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(r[e]));
        if constexpr (s == 4) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(r[e]));
        if constexpr (s == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[e]));
        if constexpr (s == 7) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(r[e]), "v"(r[(e + 1) & 7]));
    }
    if constexpr (KIND == 6) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(la));
    if constexpr (KIND == 7) asm volatile("s_nop 0");
    // LDS stores (a wave-private 4 KiB window): one every 4th / 16th MFMA is what an epilogue piece needs
    if constexpr (KIND == 8) asm volatile("ds_write_b64 %0, %1" ::"v"(la), "v"(*(const unsigned long long*)&ld));
    if constexpr (KIND == 9) { if constexpr ((N & 15) == 0) asm volatile("ds_write_b64 %0, %1" ::"v"(la), "v"(*(const unsigned long long*)&ld)); }
    if constexpr (KIND == 10) { if constexpr ((N & 15) == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(la), "v"(ld)); }
    // 16-bit forms: are the f16 transcendentals cheaper than the f32 ones, and what do packed f16 ops cost beside MFMAs?
    if constexpr (KIND == 12) asm volatile("v_exp_f16 %0, %0" : "+v"(r[e]));
    if constexpr (KIND == 13) asm volatile("v_rcp_f16 %0, %0" : "+v"(r[e]));
    if constexpr (KIND == 14) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(r[e]) : "v"(seed));
    if constexpr (KIND == 15) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(r[e]) : "v"(seed));
    if constexpr (KIND == 16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(r[e]), "v"(r[(e + 1) & 7]));
    if constexpr (KIND == 17) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r[e]) : "v"(pk));
    if constexpr (KIND == 18) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(r[e]) : "v"(seed));
    if constexpr (KIND == 11) { if constexpr ((N & 15) == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(la), "v"(ld[0])); }
}
// KIND: 0 v_fma_f32 (VOP3), 1 v_mul_f32 (VOP2), 2 v_exp_f32, 3 v_accvgpr_read (of the other half of the file), 4 v_cvt_pk_bf16_f32,
//       5 mix of the QuickGELU chain (fma fma mul exp add rcp mul cvt round robin), 6 ds_read_b128, 7 s_nop 0 (scalar issue only)
template <int KIND, int K>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    const unsigned wv = 0x3f803f80u + (unsigned)iters * 0, xv = 0x3c003c00u;
    u32x4 w, x;
    w[0] = w[1] = w[2] = w[3] = wv + (seed > 1e30f);
    x[0] = x[1] = x[2] = x[3] = xv + (seed > 1e30f);
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = seed + i + threadIdx.x * 1e-3f;
    unsigned pk = 0;
    u32x4 ld = {0, 0, 0, 0};
    const char* lp = lds + (threadIdx.x & 63) * 16;
    const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)lp;
    sfor<32>([&](auto i_) { mf<decltype(i_)::value, true>(w, x); });
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        sfor<32>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            mf<i, false>(w, x);
            sfor<K>([&](auto k_) { filler<KIND, i * K + decltype(k_)::value>(r, seed, pk, ld, la); });
        });
        if constexpr (KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n\ts_nop 7");
    float a0;
    asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(a0));
    float s = a0 + pk + ld[0];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 1234.5f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND, int K>
void run(const char* name, unsigned long long* d, int n_cu) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<KIND, K>), dim3(n_cu), dim3(256), 0, 0, d, 50, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KIND, K>), dim3(n_cu), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double mf = 32.0 * iters;
    printf("%-30s K=%d  %6.2f shader-clock ticks/MFMA   %7.2f ns/MFMA  -> %6.1f TFLOP/s chip  (ticks are 100 MHz s_memtime if < 3)\n", name, K, h[0] / mf, ms * 1e6 / mf,
           n_cu * 4 * 16384.0 / (ms * 1e6 / mf) / 1e3);
}
template <int KIND> void sweep(const char* name, unsigned long long* d, int n_cu) {
    run<KIND, 0>(name, d, n_cu); run<KIND, 1>(name, d, n_cu); run<KIND, 2>(name, d, n_cu); run<KIND, 3>(name, d, n_cu); run<KIND, 4>(name, d, n_cu);
}
int main(int argc, char** argv) {
    const int n_cu = argc > 1 ? atoi(argv[1]) : 256;
    unsigned long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    sweep<0>("v_fma_f32", d, n_cu);
    sweep<1>("v_mul_f32", d, n_cu);
    sweep<2>("v_exp_f32", d, n_cu);
    sweep<3>("accvgpr_read", d, n_cu);
    sweep<4>("cvt_pk_bf16", d, n_cu);
    sweep<5>("gelu mix", d, n_cu);
    sweep<6>("ds_read_b128", d, n_cu);
    sweep<7>("s_nop", d, n_cu);
    sweep<8>("ds_write_b64", d, n_cu);
    sweep<9>("ds_write_b64 1 per 16 fillers", d, n_cu);
    sweep<10>("ds_write_b128 1 per 16", d, n_cu);
    sweep<11>("ds_write_b32 1 per 16", d, n_cu);
    sweep<12>("v_exp_f16", d, n_cu);
    sweep<13>("v_rcp_f16", d, n_cu);
    sweep<14>("v_pk_mul_f16", d, n_cu);
    sweep<15>("v_pk_fma_f16", d, n_cu);
    sweep<16>("v_cvt_pk_f16_f32", d, n_cu);
    sweep<17>("v_cvt_f32_f16", d, n_cu);
    sweep<18>("v_pk_add_f16", d, n_cu);
    return 0;
}
