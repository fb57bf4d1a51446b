// Tuning probe: the GEMM's lean epilogue in isolation (no K loop): cycles per 256x256 tile per CU.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I mage_amd/csrc tools/probes/epi_probe.hip mage_amd/csrc/runtime.hip -o epi_probe.bin
#include "../../mage_amd/csrc/gemm.hip"
#include <cstdio>
#include <vector>

template <typename OT, int ACT>
__global__ __launch_bounds__(512, 2) void epi_test(mage_gemm_desc d, int tiles, int ntn, unsigned long long* stamps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{(float)lane, (float)a, (float)b, 1.f};
    __shared__ __attribute__((aligned(16))) char stg_all[8 * 4096];
    f32x4 biasm[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) biasm[b] = *(const f32x4*)(d.bias + wn * 64 + b * 16 + (lane >> 4) * 4);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < tiles; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const int tm = tile / ntn, tn = tile - tm * ntn;
        epilogue_lean<ACT, OT, 8>(d, biasm, acc, tm * 256 + wm * 128, tn * 256 + wn * 64, lane, d.out_h * d.out_w, stg_all + wave * 4096);
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf16x8{}, bf16x8{}, acc[a][b], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = t1 - t0;
}

int main(int argc, char** argv) {
    const int M = 262144, N = argc > 1 ? atoi(argv[1]) : 1536;
    const bool f32 = argc > 2 && atoi(argv[2]);
    mage_init(0);
    void* Y; float* bias; unsigned long long* st;
    hipMalloc(&Y, (size_t)M * N * 4); hipMalloc(&bias, N * 4); hipMemset(bias, 0, N * 4); hipMalloc(&st, 256 * 8 * 8);
    mage_gemm_desc d = {};
    d.M = M; d.N = N; d.K = 512; d.Y = Y; d.ldy = N; d.out_h = 1; d.out_w = M; d.y_mul_x = 1; d.bias = bias;
    const int ntn = N / 256, tiles = (M / 256) * ntn / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int g : {8, 64, 256}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (f32) hipLaunchKernelGGL((epi_test<float, 0>), dim3(g), dim3(512), 0, 0, d, tiles, ntn, st);
            else hipLaunchKernelGGL((epi_test<unsigned short, 0>), dim3(g), dim3(512), 0, 0, d, tiles, ntn, st);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256 * 8); hipMemcpy(h.data(), st, 256 * 8 * 8, hipMemcpyDeviceToHost);
        double mean = 0, w0 = 0, w4 = 0; for (int i = 0; i < g * 8; ++i) mean += h[i]; mean /= g * 8;
        for (int i = 0; i < g; ++i) { w0 += h[i * 8]; w4 += h[i * 8 + 4]; }
        printf("N=%d %s grid %3d: %.3f ms, %.2f us/tile/CU, shader cycles per tile (wave mean) %.0f  [wave0 %.0f wave4 %.0f], %.2f TB/s\n", N,
               f32 ? "f32" : "bf16", g, ms, ms * 1e3 / tiles, mean / tiles, w0 / g / tiles, w4 / g / tiles,
               (double)g * tiles * 65536.0 * (f32 ? 4 : 2) / ms / 1e9);
    }
    return 0;
}
