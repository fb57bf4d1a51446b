// Tuning / bring-up probe of the one-wave-per-SIMD GEMM (mage_amd/csrc/gemm4.hip) against the library's 8-wave kernels:
// bitwise comparison of the outputs and interleaved HIP-event timing on the decoder's shapes (M = 262144 rows).
// build (from the repo root; the library must be built first):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMAGE4_ABL=n] tools/probes/gemm4_probe.hip -Lmage_amd/lib -lmage_hip -Wl,-rpath,'$ORIGIN/../../mage_amd/lib' -o tools/probes/gemm4_probe.bin
#define mage_gemm4_try mage_gemm4_try_probe      // this file's own copy of the kernel (tuning builds), beside the library's
#include "../../mage_amd/csrc/gemm4.hip"
#include <cstring>
#include <vector>

static unsigned short f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float frand(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) & 0xffff) / 32768.0f - 1.0f;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 262144;
    const int rounds = argc > 2 ? atoi(argv[2]) : 5;
    if (mage_init(0) != MAGE_OK) { printf("mage_init: %s\n", mage_last_error()); return 1; }
    struct Shape { const char* name; int N, K, act; bool ln; };
    const Shape shapes[] = {{"qkv  N1536 K512 ln", 1536, 512, MAGE_ACT_NONE, true},
                            {"c_fc N2048 K512 ln+gelu", 2048, 512, MAGE_ACT_QUICKGELU, true},
                            {"plain N512 K2048", 512, 2048, MAGE_ACT_NONE, false},
                            {"plain N512 K512", 512, 512, MAGE_ACT_NONE, false},
                            {"plain N2048 K2048", 2048, 2048, MAGE_ACT_NONE, false}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int bad = 0;
    for (const Shape& sh : shapes) {
        const int N = sh.N, K = sh.K;
        std::vector<unsigned short> hA((size_t)M * K), hW((size_t)N * K);
        std::vector<float> hb(N), hcs(N), hst((size_t)M * 2);
        unsigned seed = 12345u + N * 7 + K;
        for (auto& v : hA) v = f2bf(frand(seed));
        const float ws = 1.0f / sqrtf((float)K);
        for (auto& v : hW) v = f2bf(frand(seed) * ws * 1.7f);
        for (auto& v : hb) v = frand(seed) * 0.1f;
        for (auto& v : hcs) v = frand(seed) * 0.3f;
        for (size_t i = 0; i < (size_t)M; ++i) { hst[2 * i] = frand(seed) * 0.05f; hst[2 * i + 1] = 1.0f + frand(seed) * 0.2f; }
        void *A, *W, *Y0, *Y1;
        float *b, *cs, *st;
        hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&Y0, (size_t)M * N * 2); hipMalloc(&Y1, (size_t)M * N * 2);
        hipMalloc((void**)&b, N * 4); hipMalloc((void**)&cs, N * 4); hipMalloc((void**)&st, (size_t)M * 8);
        hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
        hipMemcpy(cs, hcs.data(), N * 4, hipMemcpyHostToDevice);
        hipMemcpy(st, hst.data(), (size_t)M * 8, hipMemcpyHostToDevice);
        hipMemset(Y0, 0xff, (size_t)M * N * 2);
        hipMemset(Y1, 0xee, (size_t)M * N * 2);
        mage_gemm_desc d = {};
        d.dtype = MAGE_BF16; d.M = M; d.N = N; d.K = K; d.A = A; d.W = W; d.lda = K; d.ldy = N; d.y_dtype = MAGE_BF16;
        d.out_h = 1; d.out_w = M; d.in_h = 1; d.in_w = M; d.taps_h = d.taps_w = 1; d.cin = K; d.stride = 1; d.dys = d.dxs = 1;
        d.y_mul_x = 1; d.bias = b; d.act = sh.act; d.n_split = 1;
        if (sh.ln) { d.ln_stats = st; d.ln_colsum = cs; }
        mage_gemm_desc d0 = d, d1 = d;
        d0.Y = Y0;
        d1.Y = Y1;
        // the library's own dispatch stays on the 8-wave kernels (option gemm_no_4w = 1); this file's copy of the kernel runs with it cleared
        auto run8 = [&]() { mage_set_option("gemm_no_4w", 1); const int r_ = mage_gemm(&d0, nullptr); mage_set_option("gemm_no_4w", 0); return r_; };
        if (run8() != MAGE_OK) { printf("mage_gemm: %s\n", mage_last_error()); return 1; }
        const int r = mage_gemm4_try(&d1, nullptr);
        if (r != 1) { printf("%s: gemm4 not eligible (%d) %s\n", sh.name, r, mage_last_error()); continue; }
        if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        std::vector<unsigned short> h0((size_t)M * N), h1((size_t)M * N);
        hipMemcpy(h0.data(), Y0, h0.size() * 2, hipMemcpyDeviceToHost);
        hipMemcpy(h1.data(), Y1, h1.size() * 2, hipMemcpyDeviceToHost);
        size_t ndiff = 0, first = (size_t)-1;
        for (size_t i = 0; i < h0.size(); ++i)
            if (h0[i] != h1[i]) { if (!ndiff) first = i; ++ndiff; }
        // host reference on a few rows (fp64), so that "both wrong the same way" cannot pass
        double maxerr = 0;
        for (int rr = 0; rr < 8; ++rr) {
            const size_t m = (size_t)rr * (M / 8) + 17 * rr;
            for (int n = 0; n < N; n += 37) {
                double acc = 0;
                for (int k = 0; k < K; ++k) {
                    unsigned ua = (unsigned)hA[m * K + k] << 16, uw = (unsigned)hW[(size_t)n * K + k] << 16;
                    float fa, fw;
                    memcpy(&fa, &ua, 4); memcpy(&fw, &uw, 4);
                    acc += (double)fa * fw;
                }
                double v = sh.ln ? (acc - hst[2 * m] * hcs[n]) * hst[2 * m + 1] + hb[n] : acc + hb[n];
                if (sh.act == MAGE_ACT_QUICKGELU) v = v / (1.0 + exp(-1.702 * v));
                unsigned ug = (unsigned)h1[m * N + n] << 16;
                float got;
                memcpy(&got, &ug, 4);
                const double err = fabs(got - v) / (1.0 + fabs(v));
                if (err > maxerr) maxerr = err;
            }
        }
        printf("%-26s bitwise diffs vs 8-wave kernel: %zu of %zu (first at %zu), max rel err vs fp64 host %.3e\n", sh.name, ndiff, h0.size(), first, maxerr);
        if (ndiff || maxerr > 2e-2) ++bad;
        double t8 = 0, t4 = 0;
        for (int rd = 0; rd < rounds; ++rd) {
            float ms;
            hipEventRecord(e0);
            mage_set_option("gemm_no_4w", 1);
            for (int i = 0; i < 4; ++i) mage_gemm(&d0, nullptr);
            mage_set_option("gemm_no_4w", 0);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            t8 += ms / 4;
            hipEventRecord(e0);
            for (int i = 0; i < 4; ++i) mage_gemm4_try(&d1, nullptr);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            t4 += ms / 4;
        }
        const double fl = 2.0 * M * N * K;
        printf("%-26s 8-wave %8.1f us %7.1f TFLOP/s | 4-wave %8.1f us %7.1f TFLOP/s  (%+.1f %%)\n", sh.name, t8 / rounds * 1e3, fl / (t8 / rounds) / 1e9,
               t4 / rounds * 1e3, fl / (t4 / rounds) / 1e9, (t8 / t4 - 1) * 100);
#ifdef MAGE4_STAMP
        {
            mage_gemm4_try(&d1, nullptr);
            hipDeviceSynchronize();
            std::vector<unsigned long long> st4(256 * 32 * 4 * 8);
            hipMemcpyFromSymbol(st4.data(), HIP_SYMBOL(g4_stamps), st4.size() * 8);
            const int ntile = (M / 256) * (N / 256) / 256;
            double kl = 0, ep = 0, gap = 0, rt = 0; int cnt = 0;
            for (int b = 0; b < 256; ++b)
                for (int t = 1; t < ntile && t < 32; ++t) {
                    const unsigned long long* q = &st4[((b * 32 + t) * 4 + 0) * 8];
                    const unsigned long long* qp = &st4[((b * 32 + t - 1) * 4 + 0) * 8];
                    kl += (double)(q[1] - q[0]); ep += (double)(q[2] - q[1]); gap += (double)(q[0] - qp[2]); rt += (double)(q[3] - q[4]); ++cnt;
                }
            printf("   stamps (wave 0, tiles 1..): K loop %.0f cycles, epilogue %.0f, epilogue end -> next tile start %.0f; shader clock over the K loops %.0f MHz\n",
                   kl / cnt, ep / cnt, gap / cnt, kl / (rt / 100.0));
            // spread of the K-loop-end wall clock (100 MHz ticks) over the workgroups, per tile round
            for (int t : {1, 4, 8, 16}) {
                if (t >= ntile) break;
                unsigned long long lo = ~0ull, hi = 0;
                for (int b = 0; b < 256; ++b) { const unsigned long long v = st4[((b * 32 + t) * 4 + 0) * 8 + 3]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
                printf("   tile round %2d: K-loop-end wall clock spread over the 256 workgroups %.2f us\n", t, (hi - lo) / 100.0);
            }
        }
#endif
        hipFree(A); hipFree(W); hipFree(Y0); hipFree(Y1); hipFree(b); hipFree(cs); hipFree(st);
    }
    printf(bad ? "FAILED\n" : "OK\n");
    return bad ? 2 : 0;
}
