// conv3x3_c64_kernel: the 64 -> 64 channel 3x3 convolutions of the f8 VQ-VAE's bottleneck blocks (vqvae_model.py:126-166: hidden width dim/4 = 64;
// DecoderBlock convolutions at 32^2 .. 128^2 pixels, 2.4 of the decoder's 11.3 GFLOP per frame) as a TILE convolution: a workgroup owns 16 x 16
// output pixels, fetches their 18 x 18 x 64-channel input window ONCE into LDS (41 KB) and takes all nine taps from there; the 64 x 576 weight
// matrix (72 KB) stays in LDS for the workgroup's whole tile list.  The implicit-GEMM form of the same layer (gemm_kernel, 256 x 64 tile, one
// K slab per tap) moves 9 x 32 KB of activations + 9 x 8 KB of weights from L2 into LDS per tile -- 40 KB per 2.1 MFLOP -- and ran at 0.15-0.17
// of the matrix peak, bound by that traffic's round trips; here a tile moves 41 KB per 18.9 MFLOP.
//
// Same arithmetic, same bits: per output element the accumulation chain is the gather kernel's -- taps in (ky, kx) order, per tap two
// v_mfma_f32_16x16x32_bf16 over channels 0-31 and 32-63 with the same lane <-> k assignment, fp32 accumulators from 0, zeros for the taps
// outside the image -- and the epilogue is y = act(acc + bias) rounded to bf16 (tests compare the two kernels bitwise).
//
// One workgroup of 4 waves per CU (157.7 KB of LDS); wave w owns image rows 4w .. 4w+3 of the tile: 4 row-tiles x 4 column blocks of
// accumulators.  Per tile: barrier (the window has landed, every wave has left the previous tile) -> LDS-DMA of the NEXT tile's window into the
// other buffer -> 18 steps (tap, channel half) of 4 + 4 fragment reads and 16 MFMAs, fragments requested one step ahead -> barrier -> the
// consumed window buffer serves as the store staging area (rows leave as 8 x 128 contiguous bytes per instruction).
// a_half: the input lives at half resolution (nn.Upsample(scale_factor=2) folded into the window fetch: pixel (iy, ix) reads (iy/2, ix/2)).
#include "gemm_shared.h"

namespace {

struct Conv64Args {
    const unsigned short* A;
    const unsigned short* W;           // [64][9][64]: (cout, tap, cin)
    const float* bias;
    unsigned short* Y;
    const char* zero;
    long a_img_stride, lda;            // rows per input image, elements per input row
    long y_img_stride, y_mul_y, y_off, ldy;
    int H, Wd, a_half, relu;
    int tiles_x, tiles_per_img, ntiles;
};

constexpr int C64_W_BYTES = 9 * 8192;          // 9 taps x 64 output channels x 128 B
constexpr int C64_HALO_PIX = 18 * 18;
constexpr int C64_HALO_DMA = (C64_HALO_PIX + 7) / 8;     // 41 wave-wide DMAs of 8 pixels x 128 B
constexpr int C64_HALO_BYTES = C64_HALO_DMA * 1024;
constexpr int C64_LDS = C64_W_BYTES + 2 * C64_HALO_BYTES;
constexpr int C64_DPW = (C64_HALO_DMA + 3) / 4;          // DMAs per wave (11; wave 3 has 8)

__global__ __launch_bounds__(256) void conv3x3_c64_kernel(const Conv64Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, grp = lane >> 4;

    // ---- tile list: workgroup b runs on XCD b % 8; each XCD walks a contiguous chunk of the (image, tile row, tile column) list side by side,
    // so neighbouring tiles' shared window rows and the weights stay in that XCD's L2
    const int nwg8 = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int q8 = g.ntiles >> 3, r8 = g.ntiles & 7;
    const int chunk0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int chunk1 = chunk0 + q8 + (xcd < r8 ? 1 : 0);
    int tile = chunk0 + li;
    if (tile >= chunk1) return;

    // ---- the weights: [tap][cout][64 channels] rows of 128 B, 16-byte chunk c of row n at chunk c ^ ((n >> 1) & 7) (the GEMM kernels' image)
    for (int q = tid; q < 9 * 64 * 8; q += 256) {
        const int tap = q / 512, rem = q - tap * 512, n = rem >> 3, c = rem & 7;
        const u32x4 v = *(const u32x4*)(g.W + ((long)n * 9 + tap) * 64 + c * 8);
        *(u32x4*)(smem + tap * 8192 + n * 128 + ((c ^ ((n >> 1) & 7)) << 4)) = v;
    }

    // ---- window loader: DMA j of this wave (global DMA index dj = wave + 4 j) fills window pixels 8 dj .. 8 dj + 7; lane -> pixel 8 dj + (lane >> 3),
    // physical chunk lane & 7 = logical chunk ^ ((hx >> 1) & 7).  Per lane and DMA: window coordinates (tile-independent) and the logical chunk
    int hyx[C64_DPW];                  // hy << 8 | hx, or -1 (past the window)
    int lch[C64_DPW];
#pragma unroll
    for (int j = 0; j < C64_DPW; ++j) {
        const int p = (wave + 4 * j) * 8 + (lane >> 3);
        const int hy = p / 18, hx = p - hy * 18;
        hyx[j] = (wave + 4 * j < C64_HALO_DMA && p < C64_HALO_PIX) ? (hy << 8 | hx) : -1;
        lch[j] = ((lane & 7) ^ ((hx >> 1) & 7)) * 8;
    }
    const int in_w = g.a_half ? g.Wd >> 1 : g.Wd;
    auto issue_window = [&](int t, int buf) {
        const int img = t / g.tiles_per_img, rem = t - img * g.tiles_per_img;
        const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
        const int y0 = ty * 16 - 1, x0 = tx * 16 - 1;
        const unsigned short* base = g.A + (long)img * g.a_img_stride * g.lda;
        char* dst = smem + C64_W_BYTES + buf * C64_HALO_BYTES;
#pragma unroll
        for (int j = 0; j < C64_DPW; ++j) {
            if (wave + 4 * j < C64_HALO_DMA) {         // wave-uniform
                const int iy = y0 + (hyx[j] >> 8), ix = x0 + (hyx[j] & 255);
                const char* src = g.zero;
                if (hyx[j] >= 0 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.Wd) {
                    const long row = g.a_half ? (long)(iy >> 1) * in_w + (ix >> 1) : (long)iy * in_w + ix;
                    src = (const char*)(base + row * g.lda + lch[j]);
                }
                glds16(src, dst + (wave + 4 * j) * 1024);
            }
        }
    };
    issue_window(tile, 0);

    // ---- compute state.  Fragment of window row r (0..17), tap column kx, channel half t: pixel (r, kx + l15), chunk (grp + 4 t) ^ (((kx + l15) >> 1) & 7)
    int xoff[3][2];                    // byte offset inside a window row for (kx, t)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int t = 0; t < 2; ++t) xoff[kx][t] = (kx + l15) * 128 + (((grp + 4 * t) ^ (((kx + l15) >> 1) & 7)) << 4);
    const int rsw = (l15 >> 1) & 7;
    const int woff[2] = {l15 * 128 + ((grp ^ rsw) << 4), l15 * 128 + (((grp + 4) ^ rsw) << 4)};
    f32x4 biasv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) biasv[nt] = g.bias ? *(const f32x4*)(g.bias + nt * 16 + grp * 4) : f32x4{0.f, 0.f, 0.f, 0.f};

    __builtin_amdgcn_s_waitcnt(0x0070);                // vmcnt(0) lgkmcnt(0): the weight image is written, the first window's share has landed
    int buf = 0;
    for (bool first = true; tile < chunk1; tile += nwg8, buf ^= 1, first = false) {
        // this wave's share of the window has landed: its DMAs are older than the 8 row stores of the previous tile's epilogue, and memory
        // operations retire in order -- the stores' acknowledgements are not waited for
        if (!first) __builtin_amdgcn_s_waitcnt(0x0F78);         // vmcnt(8)
        ring_barrier();                                // the window (first tile: and the weights) is in LDS; every wave has left the other buffer
        const int next = tile + nwg8;
        if (next < chunk1) issue_window(next, buf ^ 1);
        const char* win = smem + C64_W_BYTES + buf * C64_HALO_BYTES + (wave * 4) * (18 * 128);
        f32x4 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 wf[2][4], xf[2][4];
        auto fetch = [&](int s, int slot) __attribute__((always_inline)) {        // step s = tap * 2 + t
            const int tap = s >> 1, t = s & 1, ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[slot][nt] = *(const u32x4*)(smem + tap * 8192 + nt * 2048 + woff[t]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                MAGE_DASSERT((wave * 4 + mt + ky) * (18 * 128) + xoff[kx][t] + 16 <= C64_HALO_PIX * 128 && tap * 8192 + 3 * 2048 + woff[t] + 16 <= C64_W_BYTES);
                xf[slot][mt] = *(const u32x4*)(win + (mt + ky) * (18 * 128) + xoff[kx][t]);
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            if (s + 1 < 18) fetch(s + 1, (s + 1) & 1);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt][nt] = mfma16x16x32<unsigned short>(wf[s & 1][nt], xf[s & 1][mt], acc[mt][nt]);
        }
        ring_barrier();                                // every wave is done with this window: it becomes the staging area
        // ---- epilogue: y = act(acc + bias) -> bf16; row-tile mt = image row 4 wave + mt, 16 pixels x 64 channels = 2 KB, through this wave's 2 KB
        // staging window (chunk c of row r at c ^ (r & 7)), out as 2 x (8 rows x 128 B)
        char* stg = smem + C64_W_BYTES + buf * C64_HALO_BYTES + wave * 2048;
        const int img = tile / g.tiles_per_img, rem = tile - img * g.tiles_per_img;
        const int ty = rem / g.tiles_x, tx = rem - ty * g.tiles_x;
        MAGE_DASSERT(tile >= 0 && tile < g.ntiles && ty * 16 < g.H && tx * 16 < g.Wd && 4 * 2048 <= C64_HALO_BYTES);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4 v = acc[mt][nt] + biasv[nt];
                if (g.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                const uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *(uint2*)(stg + l15 * 128 + (((nt * 2 + (grp >> 1)) ^ (l15 & 7)) << 4) + (grp & 1) * 8) = pk;
            }
            const long yrow0 = (long)img * g.y_img_stride + (long)(ty * 16 + wave * 4 + mt) * g.y_mul_y + tx * 16 + g.y_off;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = h * 8 + (lane >> 3), c = lane & 7;
                const u32x4 o = *(const u32x4*)(stg + r * 128 + ((c ^ (r & 7)) << 4));
                __builtin_nontemporal_store(o, (u32x4*)(g.Y + (yrow0 + r) * g.ldy + c * 8));
            }
        }
    }
}

}  // namespace

// 1 = launched, 0 = not eligible (mage_gemm falls through to the implicit-GEMM kernels), < 0 = error.
// Eligible: bf16 3x3 / stride 1 / pad 1 convolutions 64 -> 64 channels over whole 16 x 16 pixel tiles, epilogue act(acc + bias) with act none | ReLU,
// bf16 rows out (any row regrouping mage_gemm's y_* fields describe with y_mul_x == 1), optionally a_half; at least one tile per CU.
int mage_conv3x3_c64_try(const mage_gemm_desc* d, hipStream_t s) {
    if (mage_options().conv_no_tile) return 0;
    if (d->dtype != MAGE_BF16 || d->y_dtype != MAGE_BF16 || d->N != 64 || d->cin != 64 || d->K != 576) return 0;
    if (d->taps_h != 3 || d->taps_w != 3 || d->stride != 1 || d->dy0 != -1 || d->dx0 != -1 || d->dys != 1 || d->dxs != 1) return 0;
    if (d->in_h != d->out_h || d->in_w != d->out_w || d->out_h % 16 || d->out_w % 16 || d->out_h > 4096 || d->out_w > 4096) return 0;
    if (d->scale || d->rowadd || d->residual || d->post_relu || d->y2 || d->ln_part || d->ln_stats || d->ln_colsum || d->head_w || d->a_relu || d->n_split != 1) return 0;
    if (d->act != MAGE_ACT_NONE && d->act != MAGE_ACT_RELU) return 0;
    if (d->y_mul_x != 1 || d->ldy % 8 || d->lda % 8 || d->a_off != 0 || (d->ldw != 0 && d->ldw != d->K)) return 0;
    const long plane = (long)d->out_h * d->out_w;
    if (d->M % plane) return 0;
    const long n_img = d->M / plane;
    if ((d->y_off * d->ldy) % 8 || (d->y_mul_y * d->ldy) % 8 || (d->y_img_stride * d->ldy) % 8) return 0;
    const int dev = mage_device_index();
    if (dev < 0) return 0;
    static int n_cu_dev[MAGE_MAX_DEVICES] = {0};
    static bool attr[MAGE_MAX_DEVICES] = {false};
    if (!n_cu_dev[dev]) {
        hipDeviceProp_t p;
        n_cu_dev[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount >= 8) ? (p.multiProcessorCount & ~7) : 256;
    }
    const int n_cu = n_cu_dev[dev];
    const long ntiles = n_img * (d->out_h / 16) * (d->out_w / 16);
    if (ntiles < n_cu || ntiles >= (1L << 31) || mage_zero_page() == nullptr) return 0;
    if (!attr[dev]) {
        (void)hipFuncSetAttribute((const void*)conv3x3_c64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS);
        attr[dev] = true;
    }
    Conv64Args a;
    a.A = (const unsigned short*)d->A;
    a.W = (const unsigned short*)d->W;
    a.bias = d->bias;
    a.Y = (unsigned short*)d->Y;
    a.zero = (const char*)mage_zero_page();
    a.a_img_stride = d->a_img_stride;
    a.lda = d->lda;
    a.y_img_stride = d->y_img_stride;
    a.y_mul_y = d->y_mul_y;
    a.y_off = d->y_off;
    a.ldy = d->ldy;
    a.H = d->out_h;
    a.Wd = d->out_w;
    a.a_half = d->a_half;
    a.relu = d->act == MAGE_ACT_RELU;
    a.tiles_x = d->out_w / 16;
    a.tiles_per_img = (d->out_h / 16) * a.tiles_x;
    a.ntiles = (int)ntiles;
    hipLaunchKernelGGL(conv3x3_c64_kernel, dim3(n_cu), dim3(256), C64_LDS, s, a);
    MAGE_CHECK_LAUNCH("mage_gemm");
    return 1;
}
