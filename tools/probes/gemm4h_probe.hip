// Bring-up / tuning probe of the split-half one-wave-per-SIMD GEMM (mage_amd/csrc/gemm4h.hip; this file compiles its own copy, so the MAGE4H_*
// tuning flags work) against the library's gemm4_kernel (mage_gemm with option gemm_no_4h = 1): bitwise comparison of the outputs and interleaved HIP-event timing on the decoder's K = 512 shapes.
// build (from the repo root; the library must be built first):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMAGE4H_ABL=n] tools/probes/gemm4h_probe.hip -Lmage_amd/lib -lmage_hip -Wl,-rpath,'$ORIGIN/../../mage_amd/lib' -o tools/probes/gemm4h_probe.bin
// MAGE4H_ABL: 1 = K loop only (no outputs: the comparison is skipped), 2 = epilogue ops without global stores (likewise)
#define mage_gemm4h_try mage_gemm4h_try_probe      // this file's own copy of the kernel (tuning builds), beside the library's
#include "../../mage_amd/csrc/gemm4h.hip"
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
    const int rounds = argc > 1 ? atoi(argv[1]) : 5;
    if (mage_init(0) != MAGE_OK) { printf("mage_init: %s\n", mage_last_error()); return 1; }
    struct Shape { const char* name; int M, N, K, act; bool ln; };
    const Shape shapes[] = {{"qkv  N1536 ln", 262144, 1536, 512, MAGE_ACT_NONE, true},
                            {"c_fc N2048 ln+gelu", 262144, 2048, 512, MAGE_ACT_QUICKGELU, true},
                            {"plain N2048 bias", 262144, 2048, 512, MAGE_ACT_NONE, false},
                            {"plain N512 bias+gelu", 262144, 512, 512, MAGE_ACT_QUICKGELU, false},
                            {"plain N512 bias (out_proj's product without residual / sums)", 262144, 512, 512, MAGE_ACT_NONE, false},
                            {"c_fc M=65792 (257 row tiles)", 65792, 2048, 512, MAGE_ACT_QUICKGELU, true},
                            {"qkv  M=33024 (129 row tiles)", 33024, 1536, 512, MAGE_ACT_NONE, true},
                            {"c_fc M=16384 (incremental step, cfg2)", 16384, 2048, 512, MAGE_ACT_QUICKGELU, true},
                            {"qkv  M=16384 (incremental step, cfg2)", 16384, 1536, 512, MAGE_ACT_NONE, true},
                            {"c_fc M=8192 (incremental step, cfg4)", 8192, 2048, 512, MAGE_ACT_QUICKGELU, true},
                            {"qkv  M=8192 (incremental step, cfg4)", 8192, 1536, 512, MAGE_ACT_NONE, true},
                            {"c_fc M=4096", 4096, 2048, 512, MAGE_ACT_QUICKGELU, true},
                            {"qkv  M=4096", 4096, 1536, 512, MAGE_ACT_NONE, true},
                            {"c_fc M=2048", 2048, 2048, 512, MAGE_ACT_QUICKGELU, true},
                            {"qkv  M=2048", 2048, 1536, 512, MAGE_ACT_NONE, true},
                            {"c_fc M=1280", 1280, 2048, 512, MAGE_ACT_QUICKGELU, true},
                            {"qkv  M=1280", 1280, 1536, 512, MAGE_ACT_NONE, true},
                            {"c_fc M=24576", 24576, 2048, 512, MAGE_ACT_QUICKGELU, true},
                            {"qkv  M=49152", 49152, 1536, 512, MAGE_ACT_NONE, true},
                            {"c_fc M=32768", 32768, 2048, 512, MAGE_ACT_QUICKGELU, true},
                            {"qkv  M=32768", 32768, 1536, 512, MAGE_ACT_NONE, true}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int bad = 0;
    for (const Shape& sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K;
        std::vector<unsigned short> hA((size_t)M * K), hW((size_t)N * K);
        std::vector<float> hb(N), hcs(N), hst((size_t)M * 2);
        unsigned seed = 12345u + N * 7 + K + M;
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
        mage_set_option("gemm_no_4h", 1);           // the library's dispatch: gemm4_kernel
        if (mage_gemm(&d0, nullptr) != MAGE_OK) { printf("mage_gemm: %s\n", mage_last_error()); return 1; }
        mage_set_option("gemm_no_4h", 0);
        mage_set_option("gemm_4h_plain", 1);
        const int r = mage_gemm4h_try(&d1, nullptr, 256);
        if (r != 1) { printf("%s: gemm4h not eligible (%d) %s\n", sh.name, r, mage_last_error()); continue; }
        if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
#if !MAGE4H_ABL
        std::vector<unsigned short> h0((size_t)M * N), h1((size_t)M * N);
        hipMemcpy(h0.data(), Y0, h0.size() * 2, hipMemcpyDeviceToHost);
        hipMemcpy(h1.data(), Y1, h1.size() * 2, hipMemcpyDeviceToHost);
        size_t ndiff = 0, first = (size_t)-1;
        for (size_t i = 0; i < h0.size(); ++i)
            if (h0[i] != h1[i]) { if (!ndiff) first = i; ++ndiff; }
        printf("%-30s bitwise diffs vs gemm4_kernel: %zu of %zu (first at row %zu col %zu)\n", sh.name, ndiff, h0.size(), ndiff ? first / N : 0, ndiff ? first % N : 0);
        if (ndiff) {
            ++bad;
            // where: per 64-row x 64-column block of the first tiles
            size_t shown = 0;
            for (size_t i = 0; i < h0.size() && shown < 12; ++i)
                if (h0[i] != h1[i]) { printf("    row %zu col %zu: %04x vs %04x\n", i / N, i % N, h0[i], h1[i]); ++shown; i += 997; }
        }
#endif
        double t4 = 0, th = 0;
        for (int rd = 0; rd < rounds; ++rd) {
            float ms;
            hipEventRecord(e0);
            for (int i = 0; i < 8; ++i) mage_gemm(&d0, nullptr);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            t4 += ms / 8;
            hipEventRecord(e0);
            for (int i = 0; i < 8; ++i) mage_gemm4h_try(&d1, nullptr, 256);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            th += ms / 8;
        }
        const double fl = 2.0 * M * N * K;
        printf("%-30s gemm4 %8.1f us %7.1f TFLOP/s | gemm4h(ABL=%d) %8.1f us %7.1f TFLOP/s  (%+.1f %%)\n", sh.name, t4 / rounds * 1e3, fl / (t4 / rounds) / 1e9,
               MAGE4H_ABL, th / rounds * 1e3, fl / (th / rounds) / 1e9, (t4 / th - 1) * 100);
#ifdef MAGE4H_STAMP
        {
            mage_gemm4h_try(&d1, nullptr, 256);
            hipDeviceSynchronize();
            std::vector<unsigned long long> st4(256 * 4);
            hipMemcpyFromSymbol(st4.data(), HIP_SYMBOL(h_stamps), st4.size() * 8);
            double cyc = 0, wall = 0;
            for (int b2 = 0; b2 < 256; ++b2) { cyc += (double)(st4[b2 * 4 + 2] - st4[b2 * 4]); wall += (double)(st4[b2 * 4 + 3] - st4[b2 * 4 + 1]); }
            const double steps = (double)(M / 256) * (N / 256) / 256 * 16;
            printf("   stamps: pass loop %.0f shader cycles = %.1f us per workgroup -> %.0f MHz; %.0f cycles per 64-MFMA step (%.2f per MFMA)\n", cyc / 256, wall / 256 / 100.0,
                   cyc / (wall / 100.0), cyc / 256 / steps, cyc / 256 / steps / 64);
        }
#endif
        hipFree(A); hipFree(W); hipFree(Y0); hipFree(Y1); hipFree(b); hipFree(cs); hipFree(st);
    }
    printf(bad ? "FAILED\n" : "OK\n");
    return bad ? 2 : 0;
}
