// mage_resblock_table: the FIRST ResBlock of the f4 VQ-VAE decoder in one kernel (vqvae_model.py:111-124,180; bf16 decode).
//
// The block's input is x = relu(codebook[ids]) (the ResBlock's in-place ReLU acts on the embedding), so
//     t = relu(BN(conv3x3(x)))      = relu(bias3 + sum over the 9 taps of table[tap][ids[neighbour]])          (mage_table_conv's table sum)
//     y = relu(x + BN(conv1x1(t)))  = relu(x + ((t W1^T + b1) * scale1 + shift1))
// As separate launches this was: embedding -> zero-padded frame buffer (the residual), table sum -> t, 1x1 GEMM with the general
// epilogue: 3 launches, t written and read, x written and read.  Here a persistent workgroup (8 waves) walks 32-pixel tiles (two image
// rows of 16):
//   producer   every wave sums the table rows of 4 pixels (two pixels per load instruction: 32 lanes x 16 B = one 512-byte row of 256
//              channels; 18 loads in flight per lane), rounds to bf16 as the standalone kernel's store would, and writes the rows into an
//              XOR-swizzled LDS image [32 rows][512 B] -- the MFMA operand image of the NEXT tile, requested before this tile's MFMAs
//   1x1 conv   wave w owns output channels [32w, 32w + 32): its W1 fragments (32 x 256 bf16 = 64 VGPRs) stay in registers for the whole
//              kernel; 2 x 2 x 8 MFMAs (v_mfma_f32_16x16x32_bf16, k ascending: the GEMM kernels' operand placement and order, so the sums
//              carry the same bits as mage_gemm's)
//   epilogue   + b1, * scale1 + shift1, + x (recomputed from the fp32 codebook, rounded to bf16 as the frame buffer held it), ReLU,
//              bf16, through an LDS row image so that every store instruction writes two whole 512-byte rows
// HBM traffic: the ids in, the output rows out.  The table (2.4 MB bf16) and the codebook (0.5 MB) live in L2.
#include "common.h"

namespace {

constexpr int RT_ROWS = 32;                    // pixels per tile (two image rows of 16)
constexpr int RT_C = 256;                      // channels
constexpr int RT_TILE_BYTES = RT_ROWS * RT_C * 2;

struct ResblockTableArgs {
    const int64_t* ids;
    const unsigned short* table;               // bf16 [9][n_codes][256]
    const float* codebook;                     // fp32 [n_codes][256]
    const unsigned short* w1;                  // bf16 [256][256]
    const float* bias3;
    const float* b1;
    const float* scale1;
    const float* shift1;
    unsigned short* y;
    const char* zero;                          // >= 512 zero bytes: the "row" of a tap outside the image
    int* err;
    long ldy, y_img_stride, y_row_pitch, y_off;
    int n_tiles, H, n_codes, post_relu;
};

__device__ __forceinline__ void rt_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): this wave's LDS writes have landed before the barrier says so
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(512) void resblock_table_kernel(const ResblockTableArgs g) {
    __shared__ __attribute__((aligned(16))) char t_img[2][RT_TILE_BYTES];     // the 1x1 convolution's operand rows (bf16), double-buffered
    __shared__ __attribute__((aligned(16))) char y_img[RT_TILE_BYTES];        // output rows of the tile (bf16)
    __shared__ int id_img[2][4][18];                                          // ids of the tile's two image rows, one row and one column around
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, grp = lane >> 4;
    const int hh = lane >> 5, c8 = lane & 31;                                 // producer / row stores: which row of a pair, which 16-byte chunk
    const int tiles_per_img = g.H >> 1;

    // ---- this wave's slice of W1 and of the column vectors, once
    u32x4 wf[2][8];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wf[n][ks] = *(const u32x4*)(g.w1 + (long)(32 * wave + 16 * n + l15) * RT_C + ks * 32 + grp * 8);
    f32x4 b1v[2], s1v[2], t1v[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int co = 32 * wave + 16 * n + grp * 4;
        b1v[n] = *(const f32x4*)(g.b1 + co);
        s1v[n] = g.scale1 ? *(const f32x4*)(g.scale1 + co) : f32x4{1.f, 1.f, 1.f, 1.f};
        t1v[n] = g.scale1 ? *(const f32x4*)(g.shift1 + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 b3lo = *(const f32x4*)(g.bias3 + c8 * 8), b3hi = *(const f32x4*)(g.bias3 + c8 * 8 + 4);
    const float lo_clamp = g.post_relu ? 0.f : -INFINITY;

    // ids of a tile -> id_img[buf]: rows (2*tr - 1 .. 2*tr + 2) x columns (-1 .. 16); -1 = outside the image.  Two steps, a whole tile
    // apart: the global load (fetch_ids, into a register) and the checked write into LDS (put_ids) -- as one step the load's round trip
    // sat on the critical path of every tile (threads 0..71 waiting for it, everybody else for them at the barrier)
    auto fetch_ids = [&](int tile) -> long {
        long id = -1;
        if (tid < 72) {
            const int img = tile / tiles_per_img, tr = tile - img * tiles_per_img;
            const int r = tid / 18, c = tid - r * 18;
            const int iy = 2 * tr - 1 + r, ix = c - 1;
            if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < 16u && tile < g.n_tiles) id = g.ids[((long)img * g.H + iy) * 16 + ix];
            else id = -2;                                  // outside the image / past the last tile
        }
        return id;
    };
    auto put_ids = [&](long id, int buf) {
        if (tid < 72) {
            int v = -1;
            if (id != -2) {
                if (id < 0 || id >= g.n_codes) {           // the reference's nn.Embedding raises IndexError: reported by mage_check_device_errors
                    mage_raise(g.err, MAGE_DEVERR_EMBEDDING_ID, id, g.n_codes);
                    id = id < 0 ? 0 : g.n_codes - 1;
                }
                v = (int)id;
            }
            id_img[buf][tid / 18][tid % 18] = v;
        }
    };
    // the producer's loads: pixel pair j of this wave = tile rows 4*wave + 2*j + hh; 9 taps each
    u32x4 land[2][9];
    auto request = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = 4 * wave + 2 * j + hh, ry = p >> 4, px = p & 15;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int id = id_img[buf][ry + tap / 3][px + tap % 3];
                const char* src = id >= 0 ? (const char*)(g.table + ((long)tap * g.n_codes + id) * RT_C) : g.zero;
                land[j][tap] = *(const u32x4*)(src + c8 * 16);
            }
        }
    };
    auto widen_lo = [](unsigned u) { return __uint_as_float(u << 16); };
    auto widen_hi = [](unsigned u) { return __uint_as_float(u & 0xffff0000u); };
    // sums in mage_table_conv's order (bias first, then the taps ky, kx ascending), ReLU, bf16, into the operand image
    auto produce = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 a = b3lo, b = b3hi;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const u32x4 r = land[j][tap];
                a += f32x4{widen_lo(r[0]), widen_hi(r[0]), widen_lo(r[1]), widen_hi(r[1])};
                b += f32x4{widen_lo(r[2]), widen_hi(r[2]), widen_lo(r[3]), widen_hi(r[3])};
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = fmaxf(a[e], 0.f);
                b[e] = fmaxf(b[e], 0.f);
            }
            const int p = 4 * wave + 2 * j + hh;
            *(u32x4*)(t_img[buf] + p * 512 + ((c8 ^ (p & 15)) << 4)) =
                u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
        }
    };

    int tile = blockIdx.x;
    if (tile >= g.n_tiles) return;
    put_ids(fetch_ids(tile), 0);
    rt_barrier();
    request(0);
    put_ids(fetch_ids(tile + gridDim.x), 1);
    long ids_ahead = fetch_ids(tile + 2 * gridDim.x);
    produce(0);
    rt_barrier();

    int buf = 0;
    for (; tile < g.n_tiles; tile += gridDim.x, buf ^= 1) {
        const int nxt = tile + gridDim.x;
        const bool more = nxt < g.n_tiles;             // workgroup-uniform
        if (more) request(buf ^ 1);                    // the next tile's table rows: in flight under this tile's MFMAs and stores
        // ---- residual x = bf16(relu(codebook[id])) of this lane's rows / columns, requested before the MFMAs
        f32x4 res[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int id = id_img[buf][1 + m][1 + l15];                        // tile row 16*m + l15 = image row m of the pair, column l15
#pragma unroll
            for (int n = 0; n < 2; ++n) res[m][n] = *(const f32x4*)(g.codebook + (long)id * RT_C + 32 * wave + 16 * n + grp * 4);
        }
        // ---- 1x1 convolution of the tile
        f32x4 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            u32x4 af[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) af[m] = *(const u32x4*)(t_img[buf] + (16 * m + l15) * 512 + (((ks * 4 + grp) ^ l15) << 4));
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[n][ks]), __builtin_bit_cast(bf16x8, af[m]),
                                                                        acc[m][n], 0, 0, 0);
        }
        // ---- epilogue in the general GEMM epilogue's order: + bias, * scale + shift, + residual, ReLU; lane (l15, grp) holds row 16m + l15,
        // channels 32*wave + 16n + 4*grp + {0..3}
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                f32x4 v = acc[m][n] + b1v[n];
                v = v * s1v[n] + t1v[n];
                f32x4 r = res[m][n];
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], 0.f);
                const unsigned r01 = pack_bf16x2(r[0], r[1]), r23 = pack_bf16x2(r[2], r[3]);
                v += f32x4{widen_lo(r01), widen_hi(r01), widen_lo(r23), widen_hi(r23)};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], lo_clamp);
                const int row = 16 * m + l15, chunk = 4 * wave + 2 * n + (grp >> 1);
                *(uint2*)(y_img + row * 512 + ((chunk ^ l15) << 4) + (grp & 1) * 8) = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
        rt_barrier();                                  // y_img complete; every wave is done with t_img[buf] and with id_img[buf]
        put_ids(ids_ahead, buf);                       // the ids of the tile after next (read by the next iteration's request), fetched a tile ago
        ids_ahead = fetch_ids(nxt + 2 * gridDim.x);
        // ---- whole rows out: wave w stores tile rows 4w .. 4w + 3, two rows per instruction
        {
            const int img = tile / tiles_per_img, tr = tile - img * tiles_per_img;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int row = 4 * wave + 2 * q + hh;
                const u32x4 o = *(const u32x4*)(y_img + row * 512 + ((c8 ^ (row & 15)) << 4));
                const long yrow = (long)img * g.y_img_stride + (long)(2 * tr + (row >> 4)) * g.y_row_pitch + (row & 15) + g.y_off;
                __builtin_nontemporal_store(o, (u32x4*)(g.y + yrow * g.ldy + c8 * 8));
            }
        }
        if (more) produce(buf ^ 1);
        rt_barrier();                                  // t_img[buf ^ 1] complete; y_img read out
    }
}

// mage_resblock_rows: the tail of a ResBlock whose 3x3 convolution ran as a GEMM -- y = relu(x + BN(conv1x1(t))) with t in plain bf16
// rows and x, y in zero-padded frame buffers (vqvae_model.py:111-124: the decoder's second block).  The same MFMA stage and epilogue
// as above on 64-row tiles; the producer is two row fetches (t and x: whole 512-byte rows, two per instruction, 8 KB per wave in flight
// under the previous tile's MFMAs and stores) into LDS images.  HBM-bound: three 0.5 KB rows per pixel.  Replaces the 1x1 mage_gemm
// with the general epilogue (lockstep kernel, 32-byte residual / output pieces per lane) with the same bits out.
constexpr int RR_ROWS = 64;
constexpr int RR_IMG = RR_ROWS * RT_C * 2;

struct ResblockRowsArgs {
    const unsigned short* t;
    const unsigned short* res;
    const unsigned short* w1;
    const float* b1;
    const float* scale1;
    const float* shift1;
    unsigned short* y;
    long lda, ldr, ldy, img_stride, row_pitch, off;
    int n_tiles, hw, Wd, post_relu;
};

__global__ __launch_bounds__(512) void resblock_rows_kernel(const ResblockRowsArgs g) {
    extern __shared__ __attribute__((aligned(16))) char rr_smem[];
    char* t_img = rr_smem;
    char* r_img = rr_smem + RR_IMG;
    char* y_img = rr_smem + 2 * RR_IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, grp = lane >> 4;
    const int hh = lane >> 5, c8 = lane & 31;
    u32x4 wf[2][8];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wf[n][ks] = *(const u32x4*)(g.w1 + (long)(32 * wave + 16 * n + l15) * RT_C + ks * 32 + grp * 8);
    f32x4 b1v[2], s1v[2], t1v[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int co = 32 * wave + 16 * n + grp * 4;
        b1v[n] = *(const f32x4*)(g.b1 + co);
        s1v[n] = g.scale1 ? *(const f32x4*)(g.scale1 + co) : f32x4{1.f, 1.f, 1.f, 1.f};
        t1v[n] = g.scale1 ? *(const f32x4*)(g.shift1 + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float lo_clamp = g.post_relu ? 0.f : -INFINITY;
    auto widen_lo = [](unsigned u) { return __uint_as_float(u << 16); };
    auto widen_hi = [](unsigned u) { return __uint_as_float(u & 0xffff0000u); };
    auto frame_row = [&](long m) {                     // row of pixel m in the padded frame buffers
        const long img = m / g.hw;
        const int rem = (int)(m - img * g.hw), py = rem / g.Wd, px = rem - py * g.Wd;
        return img * g.img_stride + (long)py * g.row_pitch + px + g.off;
    };
    // this wave's rows of a tile: 8*wave + 2*j + hh
    u32x4 lt[4], lr[4];
    auto request = [&](int tile) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long m = (long)tile * RR_ROWS + 8 * wave + 2 * j + hh;
            lt[j] = *(const u32x4*)(g.t + m * g.lda + c8 * 8);
            lr[j] = *(const u32x4*)(g.res + frame_row(m) * g.ldr + c8 * 8);
        }
    };
    auto produce = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = 8 * wave + 2 * j + hh, o = p * 512 + ((c8 ^ (p & 15)) << 4);
            *(u32x4*)(t_img + o) = lt[j];
            *(u32x4*)(r_img + o) = lr[j];
        }
    };
    int tile = blockIdx.x;
    if (tile >= g.n_tiles) return;
    request(tile);
    produce();
    rt_barrier();
    for (; tile < g.n_tiles; tile += gridDim.x) {
        const int nxt = tile + gridDim.x;
        const bool more = nxt < g.n_tiles;             // workgroup-uniform
        if (more) request(nxt);
        f32x4 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            u32x4 af[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = *(const u32x4*)(t_img + (16 * m + l15) * 512 + (((ks * 4 + grp) ^ l15) << 4));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[n][ks]), __builtin_bit_cast(bf16x8, af[m]),
                                                                        acc[m][n], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int o = (16 * m + l15) * 512 + (((4 * wave + 2 * n + (grp >> 1)) ^ l15) << 4) + (grp & 1) * 8;
                const uint2 r = *(const uint2*)(r_img + o);
                f32x4 v = acc[m][n] + b1v[n];
                v = v * s1v[n] + t1v[n];
                v += f32x4{widen_lo(r.x), widen_hi(r.x), widen_lo(r.y), widen_hi(r.y)};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], lo_clamp);
                *(uint2*)(y_img + o) = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
        rt_barrier();                                  // y_img complete; every wave is done with t_img and r_img
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 8 * wave + 2 * q + hh;
            const u32x4 o = *(const u32x4*)(y_img + row * 512 + ((c8 ^ (row & 15)) << 4));
            __builtin_nontemporal_store(o, (u32x4*)(g.y + frame_row((long)tile * RR_ROWS + row) * g.ldy + c8 * 8));
        }
        if (more) produce();
        rt_barrier();                                  // the next tile's images complete; y_img read out
    }
}

}  // namespace

extern "C" int mage_resblock_rows(const void* t, int64_t lda, const void* w1, const float* b1, const float* scale1, const float* shift1,
                                  const void* residual, int64_t ldr, int32_t post_relu, void* y, int64_t ldy, int64_t n_img, int32_t H,
                                  int32_t W, int32_t C, int64_t img_stride, int64_t row_pitch, int64_t off, void* stream) {
    MAGE_CHECK_ARG(t && w1 && b1 && residual && y, "mage_resblock_rows: null pointer");
    MAGE_CHECK_ARG(n_img > 0 && H > 0 && W > 0 && C == RT_C && (n_img * H * W) % RR_ROWS == 0 && n_img * H * W / RR_ROWS < (1L << 31) &&
                       (long)H * W < (1L << 31),
                   "mage_resblock_rows: this kernel is built for C = 256 and a multiple of 64 pixels (got C=%d, %ld pixels)", C, (long)(n_img * H * W));
    MAGE_CHECK_ARG(!scale1 == !shift1, "mage_resblock_rows: scale1 and shift1 must be given together");
    MAGE_CHECK_ARG(lda >= C && ldr >= C && ldy >= C && (lda | ldr | ldy) % 8 == 0 && row_pitch >= W && img_stride >= (int64_t)(H - 1) * row_pitch + W,
                   "mage_resblock_rows: bad row geometry");
    MAGE_CHECK_ARG((((uintptr_t)t | (uintptr_t)w1 | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)b1 | (uintptr_t)scale1 | (uintptr_t)shift1) & 15) == 0,
                   "mage_resblock_rows: operands must be 16-byte aligned");
    ResblockRowsArgs a;
    a.t = (const unsigned short*)t;
    a.res = (const unsigned short*)residual;
    a.w1 = (const unsigned short*)w1;
    a.b1 = b1;
    a.scale1 = scale1;
    a.shift1 = shift1;
    a.y = (unsigned short*)y;
    a.lda = lda;
    a.ldr = ldr;
    a.ldy = ldy;
    a.img_stride = img_stride;
    a.row_pitch = row_pitch;
    a.off = off;
    a.n_tiles = (int)(n_img * H * W / RR_ROWS);
    a.hw = H * W;
    a.Wd = W;
    a.post_relu = post_relu;
    const int dev = mage_device_index();
    MAGE_CHECK_ARG(dev >= 0, "mage_resblock_rows: no current device");
    static int n_cu_dev[MAGE_MAX_DEVICES] = {0};
    if (!n_cu_dev[dev]) {
        hipDeviceProp_t p;
        n_cu_dev[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
        (void)hipFuncSetAttribute((const void*)resblock_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * RR_IMG);
    }
    const int grid = a.n_tiles < n_cu_dev[dev] ? a.n_tiles : n_cu_dev[dev];
    hipLaunchKernelGGL(resblock_rows_kernel, dim3(grid), dim3(512), 3 * RR_IMG, (hipStream_t)stream, a);
    MAGE_CHECK_LAUNCH("mage_resblock_rows");
    return MAGE_OK;
}

extern "C" int mage_resblock_table(const int64_t* ids, int64_t n_img, int32_t H, int32_t W, const void* table, int32_t n_codes, int32_t C,
                                   const float* bias3, const float* codebook, const void* w1, const float* b1, const float* scale1,
                                   const float* shift1, int32_t post_relu, void* y, int64_t ldy, int64_t y_img_stride, int64_t y_row_pitch,
                                   int64_t y_off, void* stream) {
    MAGE_CHECK_ARG(ids && table && bias3 && codebook && w1 && b1 && y, "mage_resblock_table: null pointer");
    MAGE_CHECK_ARG(n_img > 0 && W == 16 && H >= 2 && H % 2 == 0 && C == RT_C && n_codes > 0 && n_img * (H / 2) < (1L << 31),
                   "mage_resblock_table: this kernel is built for W = 16, even H, C = 256 (got H=%d W=%d C=%d)", H, W, C);
    MAGE_CHECK_ARG(!scale1 == !shift1, "mage_resblock_table: scale1 and shift1 must be given together");
    MAGE_CHECK_ARG(ldy >= C && ldy % 8 == 0 && y_row_pitch >= W && y_img_stride >= (int64_t)(H - 1) * y_row_pitch + W,
                   "mage_resblock_table: bad output geometry");
    MAGE_CHECK_ARG((((uintptr_t)table | (uintptr_t)w1 | (uintptr_t)y | (uintptr_t)codebook | (uintptr_t)bias3 | (uintptr_t)b1 |
                     (uintptr_t)scale1 | (uintptr_t)shift1) & 15) == 0, "mage_resblock_table: operands must be 16-byte aligned");
    ResblockTableArgs a;
    a.ids = ids;
    a.table = (const unsigned short*)table;
    a.codebook = codebook;
    a.w1 = (const unsigned short*)w1;
    a.bias3 = bias3;
    a.b1 = b1;
    a.scale1 = scale1;
    a.shift1 = shift1;
    a.y = (unsigned short*)y;
    a.zero = (const char*)mage_zero_page();
    a.err = mage_error_word();
    MAGE_CHECK_ARG(a.zero && a.err, "mage_resblock_table: mage_init() has not been called");
    a.ldy = ldy;
    a.y_img_stride = y_img_stride;
    a.y_row_pitch = y_row_pitch;
    a.y_off = y_off;
    a.n_tiles = (int)(n_img * (H / 2));
    a.H = H;
    a.n_codes = n_codes;
    a.post_relu = post_relu;
    const int dev = mage_device_index();
    MAGE_CHECK_ARG(dev >= 0, "mage_resblock_table: no current device");
    static int n_cu_dev[MAGE_MAX_DEVICES] = {0};
    if (!n_cu_dev[dev]) {
        hipDeviceProp_t p;
        n_cu_dev[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    const int grid = a.n_tiles < n_cu_dev[dev] ? a.n_tiles : n_cu_dev[dev];
    hipLaunchKernelGGL(resblock_table_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
    MAGE_CHECK_LAUNCH("mage_resblock_table");
    return MAGE_OK;
}
