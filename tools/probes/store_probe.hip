// Tuning probe: per-CU vs chip-wide limits of the GEMM epilogue store pattern (wave-store = 8 rows x 128 B vs 2 rows x 512 B, nt vs plain, 1..256 CUs).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/store_probe.hip -o store_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// each WG (512 thr = 8 waves) writes `tiles` tiles of 256 rows x 512 B (bf16 256x256) into a [M, ldy] matrix
// MODE 0: wave-store = 8 rows x 128 B (GEMM register epilogue);  MODE 1: wave-store = 2 rows x 512 B
template <int MODE, bool NT>
__global__ __launch_bounds__(512) void k(char* Y, long ldy_bytes, int tiles, int tiles_n, int total_tiles) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 2, wn = wave & 3;
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int t = 0; t < tiles; ++t) {
        int tile = (blockIdx.x + t * gridDim.x) % total_tiles;
        int tm = tile / tiles_n, tn = tile % tiles_n;
        char* base = Y + (long)tm * 256 * ldy_bytes + tn * 512;
        if (MODE == 0) {
#pragma unroll 4
            for (int r = 0; r < 16; ++r) {
                int row = wm * 128 + r * 8 + (lane >> 3);
                char* p = base + (long)row * ldy_bytes + wn * 128 + (lane & 7) * 16;
                if (NT) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
            }
        } else {
#pragma unroll 4
            for (int r = 0; r < 16; ++r) {
                int row = wave * 32 + r * 2 + (lane >> 5);
                char* p = base + (long)row * ldy_bytes + (lane & 31) * 16;
                if (NT) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
            }
        }
    }
}
int main() {
    const long M = 262144, N = 1536; const long ldy = N * 2;
    char* Y; hipMalloc(&Y, M * ldy);
    const int tiles_n = N / 256, total = (M / 256) * tiles_n;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grids[] = {1, 8, 32, 64, 128, 256};
    for (int mode = 0; mode < 2; ++mode) for (int nt = 0; nt < 2; ++nt) for (int g : grids) {
        int tiles = 24;
        auto launch = [&]() {
            if (mode == 0 && nt == 0) hipLaunchKernelGGL((k<0, false>), dim3(g), dim3(512), 0, 0, Y, ldy, tiles, tiles_n, total);
            if (mode == 0 && nt == 1) hipLaunchKernelGGL((k<0, true>), dim3(g), dim3(512), 0, 0, Y, ldy, tiles, tiles_n, total);
            if (mode == 1 && nt == 0) hipLaunchKernelGGL((k<1, false>), dim3(g), dim3(512), 0, 0, Y, ldy, tiles, tiles_n, total);
            if (mode == 1 && nt == 1) hipLaunchKernelGGL((k<1, true>), dim3(g), dim3(512), 0, 0, Y, ldy, tiles, tiles_n, total);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        double bytes = (double)g * tiles * 131072.0;
        double us_tile = ms * 1e3 / tiles;
        printf("mode %d nt %d grid %3d: %.3f ms  %.2f us/tile/CU  %.0f cyc/wave-store (2.4GHz, 128 per tile)  %.2f TB/s\n", mode, nt, g, ms,
               us_tile, us_tile * 2400 / 128, bytes / ms / 1e9);
    }
    return 0;
}
