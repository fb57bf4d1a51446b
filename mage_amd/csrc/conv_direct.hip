// Small-channel ends of the VQ-VAE (image <-> channels-last features) and the channels-last elementwise
// helpers of the f8 stack.  These are HBM/L2-bound integer-ish data movement, not GEMMs: coalesced
// channel-fastest accesses, no matrix cores.
#include "common.h"

namespace {

// y[n, oy, ox, co] = act((bias[co] + sum_{ci,ky,kx} x[n, ci, oy*s-p+ky, ox*s-p+kx] * wt[(ci,ky,kx), co]) * scale + shift)
// One thread per (pixel, 4 consecutive co); the KH*KW*cin input taps are wave-uniform broadcast loads.
template <typename OT>
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                      const float* __restrict__ bias, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, OT* __restrict__ y, int N, int cin,
                                                      int H, int W, int cout, int kh, int kw, int stride, int pad, int OH,
                                                      int OW, int act, int s2d) {
    const int cq = cout / 4;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)N * OH * OW * cq;
    if (gid >= total) return;
    const int co = (int)(gid % cq) * 4;
    const long pix = gid / cq;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((long)OW * OH));
    f32x4 acc = bias ? *(const f32x4*)(bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ci = 0; ci < cin; ++ci) {
        const float* xp = x + ((long)n * cin + ci) * H * W;
        for (int ky = 0; ky < kh; ++ky) {
            const int iy = oy * stride - pad + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < kw; ++kx) {
                const int ix = ox * stride - pad + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const float xv = xp[(long)iy * W + ix];
                acc += xv * *(const f32x4*)(wt + ((long)(ci * kh + ky) * kw + kx) * cout + co);
            }
        }
    }
    if (scale) acc = acc * *(const f32x4*)(scale + co) + *(const f32x4*)(shift + co);
    if (act == MAGE_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.f);
    }
    if (s2d) {
        // offset space-to-depth rows for a following 4x4 / stride-2 / pad-1 convolution: block (R, C) = ((oy+1)/2, (ox+1)/2) of a
        // (OH/2+1) x (OW/2+1) grid holds the 2x2 pixels (2R-1..2R, 2C-1..2C) as 4*cout channels, so that convolution is a 2x2 /
        // stride-1 window over blocks (the padded-taps form of mage_gemm); border sub-blocks are never written: they stay zero
        const int BW = OW / 2 + 1, BH = OH / 2 + 1;
        const long row = (long)n * (BH * BW) + (long)((oy + 1) >> 1) * BW + ((ox + 1) >> 1);
        const int q = ((oy + 1) & 1) * 2 + ((ox + 1) & 1);
        store4(y + row * (4 * cout) + q * cout + co, acc);
    } else {
        store4(y + pix * cout + co, acc);
    }
}

// The f4 encoder's stem, Conv2d(1, dim, 4, 2, 1) (vqvae_model.py:170), at many frames.  The kernel above re-reads the [taps][cout] weights
// from L1 for every pixel (16 KB per pixel-wave at dim = 256) and decodes and bounds-tests every tap per thread: 1.1 TB/s of output at 1024
// frames.  Here workgroup = image: the image sits in LDS with a one-pixel zero border (so no tap is ever outside: (2 oy + ky, 2 ox + kx)
// of the padded plane), a wave keeps its 4 channels x 16 taps per lane in registers and walks output pixels; a pixel's 16 taps are 8
// ds_read2_b32 off ONE address register (row pitch W + 2 <= 84 words: every tap offset fits the instruction's 8-bit field), the same
// address in every lane of a pixel (a broadcast).  Same products in the same order: the same values (acc + 0 * w where the other kernel
// skips a border tap: a -0 / +0 is the only possible difference).
template <typename OT>
__global__ __launch_bounds__(256) void conv_in_4x4s2_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                            const float* __restrict__ bias, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, OT* __restrict__ y, int H, int W, int cout,
                                                            int act, int s2d) {
    extern __shared__ float img[];                      // [(H + 2)][(W + 2)]
    const int n = blockIdx.x, PW = W + 2, OH = H / 2, OW = W / 2;
    for (int i = threadIdx.x; i < (H + 2) * PW; i += 256) img[i] = 0.f;
    __syncthreads();
    const float* xp = x + (long)n * H * W;
    for (int i = threadIdx.x * 4; i < H * W; i += 1024) {              // W % 4 == 0 (host): a 16-byte piece stays inside a row
        const f32x4 v = *(const f32x4*)(xp + i);
        const int r = i / W, c = i - r * W;
        float* d = img + (r + 1) * PW + c + 1;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    const int cq = cout / 4;                            // lanes per pixel (host: cq divides 64; 64 at cout = 256)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ppw = 64 / cq, sub = lane / cq, co = (lane % cq) * 4;
    f32x4 w[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) w[t] = *(const f32x4*)(wt + (long)t * cout + co);
    const f32x4 b4 = bias ? *(const f32x4*)(bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 s4 = scale ? *(const f32x4*)(scale + co) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 t4 = scale ? *(const f32x4*)(shift + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const int npx = OH * OW;
    for (int p = wave * ppw + sub; p < npx; p += 4 * ppw) {
        const int oy = p / OW, ox = p - oy * OW;
        const float* tp = img + (2 * oy) * PW + 2 * ox;
        f32x4 acc = b4;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) acc += tp[ky * PW + kx] * w[ky * 4 + kx];
        if (scale) acc = acc * s4 + t4;
        if (act == MAGE_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.f);
        }
        if (s2d) {
            const int BW = OW / 2 + 1, BH = OH / 2 + 1;
            const long row = (long)n * (BH * BW) + ((oy + 1) >> 1) * BW + ((ox + 1) >> 1);
            const int q = ((oy + 1) & 1) * 2 + ((ox + 1) & 1);
            store4(y + row * (4 * cout) + q * cout + co, acc);
        } else {
            store4(y + ((long)n * npx + p) * cout + co, acc);
        }
    }
}

// The 1x1 head (the f8 decoder's Conv2d(dim, C, 1) + Tanh at full resolution: 16 M pixels x 512 B per call at cfg4): 16 lanes per pixel,
// 4 pixels per wave, every lane reads 16-byte chunks (a wave instruction covers four contiguous 256-byte runs), 4 xor-shuffles per
// output channel.  The wave-per-pixel kernel below spent its time in 6-step reductions: 7.8 ms = 1.06 TB/s on that call.
template <typename IT>
__global__ __launch_bounds__(256) void conv_out_1x1_kernel(const IT* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
                                                           float* __restrict__ y, long npix, long plane, int cin, int cout) {
    constexpr int CPL = 16 / (int)sizeof(IT);                     // channels per 16-byte chunk: 8 (bf16) | 4 (fp32)
    const int sub = threadIdx.x & 15;
    const long pix = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool live = pix < npix;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const IT* xp = x + pix * cin;
        for (int c = sub * CPL; c < cin; c += 16 * CPL) {
            float xv[CPL];
#pragma unroll
            for (int q = 0; q < CPL / 4; ++q) {
                const f32x4 v = load4(xp + c + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[4 * q + e] = v[e];
            }
#pragma unroll
            for (int co = 0; co < 4; ++co) {
                if (co < cout) {
                    const float* wp = wt + (long)co * cin + c;
#pragma unroll
                    for (int q = 0; q < CPL / 4; ++q) {
                        const f32x4 wv = *(const f32x4*)(wp + 4 * q);
                        acc[co] += xv[4 * q] * wv[0] + xv[4 * q + 1] * wv[1] + xv[4 * q + 2] * wv[2] + xv[4 * q + 3] * wv[3];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int co = 0; co < 4; ++co) {
        if (co < cout) {
            float v = acc[co];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            if (live && sub == 0) {
                const long n = pix / plane, p = pix - n * plane;
                y[(n * cout + co) * plane + p] = tanhf(v + (bias ? bias[co] : 0.f));
            }
        }
    }
}

// The same head over FOUR fp32 input channels (the f8 decoder's RGB head already taken on the last convolution's tile, mage_gemm_desc::head_w
// with ldy == 4: x holds the head's sums): a thread per pixel, one 16-byte load, cout <= 4 coalesced plane stores.
__global__ __launch_bounds__(256) void conv_out_1x1_c4_kernel(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
                                                              float* __restrict__ y, long npix, long plane, int cout) {
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    const f32x4 v = __builtin_nontemporal_load((const f32x4*)(x + pix * 4));
    const long n = pix / plane, p = pix - n * plane;
    for (int co = 0; co < cout; ++co) {
        const f32x4 w = *(const f32x4*)(wt + co * 4);
        // (the 16-lane kernel's order for cin = 4: one lane holds the pixel, its four products added left to right)
        const float a = v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
        y[(n * cout + co) * plane + p] = tanhf(a + (bias ? bias[co] : 0.f));
    }
}

// One wave per output pixel, lanes split the input channels, shuffle reduction, tanh, NCHW fp32 store.
template <typename IT, bool TRANSPOSED>
__global__ __launch_bounds__(256) void conv_out_kernel(const IT* __restrict__ x, const float* __restrict__ wt,
                                                       const float* __restrict__ bias, float* __restrict__ y, int N,
                                                       int IH, int IW, int cin, int cout, int OH, int OW) {
    const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= (long)N * OH * OW) return;
    const int lane = threadIdx.x & 63;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((long)OW * OH));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (TRANSPOSED) {
        // oy = 2*iy - 1 + ky: the two contributing input rows are iy0 = (oy+1)>>1 (ky = oy+1-2*iy0) and iy0-1 (ky+2)
        const int iy0 = (oy + 1) >> 1, ky0 = oy + 1 - 2 * iy0;
        const int ix0 = (ox + 1) >> 1, kx0 = ox + 1 - 2 * ix0;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int iy = iy0 - a, ky = ky0 + 2 * a;
            if ((unsigned)iy >= (unsigned)IH) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ix = ix0 - b, kx = kx0 + 2 * b;
                if ((unsigned)ix >= (unsigned)IW) continue;
                const IT* xp = x + (((long)n * IH + iy) * IW + ix) * cin;
                for (int c = lane * 4; c < cin; c += 256) {
                    const f32x4 xv = load4(xp + c);
#pragma unroll
                    for (int co = 0; co < 4; ++co) {
                        if (co < cout) {
                            const f32x4 wv = *(const f32x4*)(wt + ((long)((ky * 4 + kx) * cout + co)) * cin + c);
                            acc[co] += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
                        }
                    }
                }
            }
        }
    } else {
        const IT* xp = x + (((long)n * IH + oy) * IW + ox) * cin;
        for (int c = lane * 4; c < cin; c += 256) {
            const f32x4 xv = load4(xp + c);
#pragma unroll
            for (int co = 0; co < 4; ++co) {
                if (co < cout) {
                    const f32x4 wv = *(const f32x4*)(wt + (long)co * cin + c);
                    acc[co] += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
                }
            }
        }
    }
#pragma unroll
    for (int co = 0; co < 4; ++co) {
        if (co < cout) {
            const float s = wave_sum(acc[co]);
            if (lane == 0) y[(((long)n * cout + co) * OH + oy) * OW + ox] = tanhf(s + (bias ? bias[co] : 0.f));
        }
    }
}

// Fold of a 4x4 / stride 2 / pad 1 transposed convolution (see mage_convt_fold_tanh): thread = output pixel (NCHW, x fastest).
__global__ __launch_bounds__(256) void convt_fold_tanh_kernel(const float* __restrict__ taps, const float* __restrict__ bias,
                                                              float* __restrict__ y, int N, int IH, int IW, int cout) {
    const int OH = IH * 2, OW = IW * 2;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)N * cout * OH * OW) return;
    const int ox = (int)(gid % OW), oy = (int)((gid / OW) % OH), co = (int)((gid / ((long)OW * OH)) % cout);
    const int n = (int)(gid / ((long)OW * OH * cout));
    // oy = 2*iy - 1 + ky: the two contributing input rows are iy0 = (oy+1)>>1 with ky0 = oy+1-2*iy0, and iy0-1 with ky0+2
    const int iy0 = (oy + 1) >> 1, ky0 = oy + 1 - 2 * iy0;
    const int ix0 = (ox + 1) >> 1, kx0 = ox + 1 - 2 * ix0;
    const int ld = 16 * cout;
    float s = bias ? bias[co] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int iy = iy0 - a, ky = ky0 + 2 * a;
        if ((unsigned)iy >= (unsigned)IH) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ix = ix0 - b, kx = kx0 + 2 * b;
            if ((unsigned)ix >= (unsigned)IW) continue;
            s += taps[(((long)n * IH + iy) * IW + ix) * ld + (ky * 4 + kx) * cout + co];
        }
    }
    y[gid] = tanhf(s);
}

// The same fold for ONE output channel and a tap plane that fits LDS (IH*IW <= 1024: the f4 decoder's 32 x 32 grid, 64 KB of taps per
// image): workgroup = image.  The per-pixel kernel above reads 4 bytes from each of 4 different 64-byte tap rows per thread (2 TB/s); here
// the image's taps come in as whole rows (16 B per lane, contiguous), sit in LDS at a pitch of 17 floats (bank = (17 p + tap) mod 64: the
// lanes of an output row hit 64 different banks), and the output rows leave contiguous.  Same sum order, same tanhf: same bits.
__global__ __launch_bounds__(256) void convt_fold_tanh_img_kernel(const float* __restrict__ taps, const float* __restrict__ bias,
                                                                  float* __restrict__ y, int IH, int IW) {
    extern __shared__ float tl[];                      // [IH*IW][17]
    const int n = blockIdx.x, npx = IH * IW;
    const f32x4* src = (const f32x4*)(taps + (long)n * npx * 16);
    for (int base = 0; base < npx * 4; base += 256 * 16) {          // 16 loads in flight per thread (a loop of dependent round trips otherwise)
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[min(base + u * 256 + (int)threadIdx.x, npx * 4 - 1)];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = base + u * 256 + threadIdx.x;
            if (i < npx * 4) {
                float* d = tl + (i >> 2) * 17 + (i & 3) * 4;
                d[0] = v[u][0]; d[1] = v[u][1]; d[2] = v[u][2]; d[3] = v[u][3];
            }
        }
    }
    __syncthreads();
    const int OH = IH * 2, OW = IW * 2;
    const float b0 = bias ? bias[0] : 0.f;
    float* yo = y + (long)n * OH * OW;
    for (int o = threadIdx.x; o < OH * OW; o += 256) {
        const int oy = o / OW, ox = o - oy * OW;
        const int iy0 = (oy + 1) >> 1, ky0 = oy + 1 - 2 * iy0;
        const int ix0 = (ox + 1) >> 1, kx0 = ox + 1 - 2 * ix0;
        float s = b0;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int iy = iy0 - a, ky = ky0 + 2 * a;
            if ((unsigned)iy >= (unsigned)IH) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ix = ix0 - b, kx = kx0 + 2 * b;
                if ((unsigned)ix >= (unsigned)IW) continue;
                s += tl[(iy * IW + ix) * 17 + ky * 4 + kx];
            }
        }
        yo[o] = tanhf(s);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool2_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W,
                                                       int C, int relu) {
    const int cq = C / 4, OH = H / 2, OW = W / 2;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)N * OH * OW * cq) return;
    const int c = (int)(gid % cq) * 4;
    const long pix = gid / cq;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((long)OW * OH));
    const T* p = x + (((long)n * H + oy * 2) * W + ox * 2) * C + c;
    f32x4 a = load4(p), b = load4(p + C), cc = load4(p + (long)W * C), dd = load4(p + (long)W * C + C), o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(cc[e], dd[e]));
        if (relu) o[e] = fmaxf(o[e], 0.f);
    }
    store4(y + pix * C + c, o);
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C) {
    const int cq = C / 4, OH = H * 2, OW = W * 2;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)N * OH * OW * cq) return;
    const int c = (int)(gid % cq) * 4;
    const long pix = gid / cq;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((long)OW * OH));
    store4(y + pix * C + c, load4(x + (((long)n * H + (oy >> 1)) * W + (ox >> 1)) * C + c));
}

template <typename IT, typename OT, bool RELU>
__global__ __launch_bounds__(256) void map_kernel(const IT* __restrict__ x, OT* __restrict__ y, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 v = load4(x + i);
    if (RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    store4(y + i, v);
}

inline dim3 grid1(long items) { return dim3((unsigned)((items + 255) / 256)); }

}  // namespace

extern "C" int mage_conv_in(const float* x, const float* weight_t, const float* bias, const float* scale, const float* shift,
                            void* y, int32_t y_dtype, int32_t N, int32_t cin, int32_t H, int32_t W, int32_t cout, int32_t kh,
                            int32_t kw, int32_t stride, int32_t pad, int32_t act, int32_t s2d, void* stream) {
    MAGE_CHECK_ARG(x && weight_t && y, "mage_conv_in: null pointer");
    MAGE_CHECK_ARG(N > 0 && cin > 0 && cin <= 4 && cout % 4 == 0 && stride >= 1, "mage_conv_in: cin=%d cout=%d unsupported", cin, cout);
    MAGE_CHECK_ARG(!scale == !shift, "mage_conv_in: scale and shift must be given together");
    MAGE_CHECK_ARG(act == MAGE_ACT_NONE || act == MAGE_ACT_RELU, "mage_conv_in: act %d unsupported", act);
    const int OH = (H + 2 * pad - kh) / stride + 1, OW = (W + 2 * pad - kw) / stride + 1;
    const long items = (long)N * OH * OW * (cout / 4);
    hipStream_t s = (hipStream_t)stream;
    MAGE_CHECK_ARG(!s2d || (OH % 2 == 0 && OW % 2 == 0), "mage_conv_in: the space-to-depth output needs an even output plane");
    const int cq = cout / 4;
    if (cin == 1 && kh == 4 && kw == 4 && stride == 2 && pad == 1 && N >= 64 && H % 2 == 0 && W % 4 == 0 && W <= 80 && (H + 2) * (W + 2) * 4 <= 48 * 1024 &&
        cq <= 64 && 64 % cq == 0 &&
        (y_dtype == MAGE_F32 || ((y_dtype == MAGE_BF16X3 || y_dtype == MAGE_F16X3) && cout % 64 == 0 && (((uintptr_t)y) & 255) == 0))) {
        // one image per workgroup through LDS, weights in registers (conv_in_4x4s2_kernel)
        const dim3 grid(N), blk(256);
        const size_t lds = (size_t)(H + 2) * (W + 2) * 4;
#define CI16(T_) hipLaunchKernelGGL((conv_in_4x4s2_kernel<T_>), grid, blk, lds, s, x, weight_t, bias, scale, shift, (T_*)y, H, W, cout, act, s2d)
        if (y_dtype == MAGE_F32) CI16(float);
        else if (y_dtype == MAGE_BF16X3) CI16(split_bf16);
        else CI16(split_f16);
#undef CI16
        MAGE_CHECK_LAUNCH("mage_conv_in");
        return MAGE_OK;
    }
    if (y_dtype == MAGE_F32)
        hipLaunchKernelGGL((conv_in_kernel<float>), grid1(items), dim3(256), 0, s, x, weight_t, bias, scale, shift, (float*)y, N,
                           cin, H, W, cout, kh, kw, stride, pad, OH, OW, act, s2d);
    else if (y_dtype == MAGE_BF16)
        hipLaunchKernelGGL((conv_in_kernel<unsigned short>), grid1(items), dim3(256), 0, s, x, weight_t, bias, scale, shift,
                           (unsigned short*)y, N, cin, H, W, cout, kh, kw, stride, pad, OH, OW, act, s2d);
    else if (y_dtype == MAGE_BF16X3 || y_dtype == MAGE_F16X3) {
        MAGE_CHECK_ARG(cout % 64 == 0 && (((uintptr_t)y) & 255) == 0, "mage_conv_in: split output needs cout %% 64 == 0 and y 256-byte aligned");
        if (y_dtype == MAGE_BF16X3)
            hipLaunchKernelGGL((conv_in_kernel<split_bf16>), grid1(items), dim3(256), 0, s, x, weight_t, bias, scale, shift, (split_bf16*)y, N,
                               cin, H, W, cout, kh, kw, stride, pad, OH, OW, act, s2d);
        else
            hipLaunchKernelGGL((conv_in_kernel<split_f16>), grid1(items), dim3(256), 0, s, x, weight_t, bias, scale, shift, (split_f16*)y, N,
                               cin, H, W, cout, kh, kw, stride, pad, OH, OW, act, s2d);
    } else {
        mage_set_error("mage_conv_in: bad y_dtype %d", y_dtype);
        return MAGE_EINVAL;
    }
    MAGE_CHECK_LAUNCH("mage_conv_in");
    return MAGE_OK;
}

extern "C" int mage_conv_out(const void* x, int32_t x_dtype, const float* weight_t, const float* bias, float* y, int32_t N,
                             int32_t IH, int32_t IW, int32_t cin, int32_t cout, int32_t transposed, void* stream) {
    MAGE_CHECK_ARG(x && weight_t && y, "mage_conv_out: null pointer");
    MAGE_CHECK_ARG(N > 0 && cin % 4 == 0 && cout >= 1 && cout <= 4, "mage_conv_out: cin=%d cout=%d unsupported", cin, cout);
    const int OH = transposed ? IH * 2 : IH, OW = transposed ? IW * 2 : IW;
    const dim3 grid((unsigned)(((long)N * OH * OW + 3) / 4)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (!transposed && (x_dtype == MAGE_F32 || x_dtype == MAGE_BF16) && cin % (x_dtype == MAGE_BF16 ? 8 : 4) == 0 &&
        (((uintptr_t)x | (uintptr_t)weight_t) & 15) == 0) {
        const long npix = (long)N * IH * IW;
        if (x_dtype == MAGE_F32 && cin == 4) {
            hipLaunchKernelGGL(conv_out_1x1_c4_kernel, dim3((unsigned)((npix + 255) / 256)), blk, 0, s, (const float*)x, weight_t, bias, y, npix, (long)IH * IW, cout);
            MAGE_CHECK_LAUNCH("mage_conv_out");
            return MAGE_OK;
        }
        const dim3 g1((unsigned)((npix * 16 + 255) / 256));
        if (x_dtype == MAGE_F32)
            hipLaunchKernelGGL((conv_out_1x1_kernel<float>), g1, blk, 0, s, (const float*)x, weight_t, bias, y, npix, (long)IH * IW, cin, cout);
        else
            hipLaunchKernelGGL((conv_out_1x1_kernel<unsigned short>), g1, blk, 0, s, (const unsigned short*)x, weight_t, bias, y, npix,
                               (long)IH * IW, cin, cout);
        MAGE_CHECK_LAUNCH("mage_conv_out");
        return MAGE_OK;
    }
#define CO_LAUNCH(IT, TR) hipLaunchKernelGGL((conv_out_kernel<IT, TR>), grid, blk, 0, s, (const IT*)x, weight_t, bias, y, N, IH, IW, cin, cout, OH, OW)
    if (x_dtype == MAGE_F32) { if (transposed) CO_LAUNCH(float, true); else CO_LAUNCH(float, false); }
    else if (x_dtype == MAGE_BF16) { if (transposed) CO_LAUNCH(unsigned short, true); else CO_LAUNCH(unsigned short, false); }
    else { mage_set_error("mage_conv_out: bad x_dtype %d", x_dtype); return MAGE_EINVAL; }
#undef CO_LAUNCH
    MAGE_CHECK_LAUNCH("mage_conv_out");
    return MAGE_OK;
}

extern "C" int mage_convt_fold_tanh(const float* taps, const float* bias, float* y, int32_t N, int32_t IH, int32_t IW,
                                    int32_t cout, void* stream) {
    MAGE_CHECK_ARG(taps && y, "mage_convt_fold_tanh: null pointer");
    MAGE_CHECK_ARG(N > 0 && IH > 0 && IW > 0 && cout >= 1 && cout <= 4, "mage_convt_fold_tanh: bad shape N=%d IH=%d IW=%d cout=%d", N,
                   IH, IW, cout);
    if (cout == 1 && IH * IW <= 1024 && N >= 64) {         // one image per workgroup through LDS (same bits)
        const int dev = mage_device_index();
        MAGE_CHECK_ARG(dev >= 0, "mage_convt_fold_tanh: no current device");
        static bool attr[MAGE_MAX_DEVICES] = {false};
        if (!attr[dev]) {
            (void)hipFuncSetAttribute((const void*)convt_fold_tanh_img_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * 17 * 4);
            attr[dev] = true;
        }
        hipLaunchKernelGGL(convt_fold_tanh_img_kernel, dim3(N), dim3(256), (size_t)IH * IW * 17 * 4, (hipStream_t)stream, taps, bias, y, IH, IW);
        MAGE_CHECK_LAUNCH("mage_convt_fold_tanh");
        return MAGE_OK;
    }
    const long items = (long)N * cout * IH * 2 * IW * 2;
    hipLaunchKernelGGL(convt_fold_tanh_kernel, grid1(items), dim3(256), 0, (hipStream_t)stream, taps, bias, y, N, IH, IW, cout);
    MAGE_CHECK_LAUNCH("mage_convt_fold_tanh");
    return MAGE_OK;
}

extern "C" int mage_maxpool2(const void* x, void* y, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t relu,
                             void* stream) {
    MAGE_CHECK_ARG(x && y && N > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "mage_maxpool2: bad arguments");
    const long items = (long)N * (H / 2) * (W / 2) * (C / 4);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MAGE_F32) hipLaunchKernelGGL((maxpool2_kernel<float>), grid1(items), dim3(256), 0, s, (const float*)x, (float*)y, N, H, W, C, relu);
    else if (dtype == MAGE_BF16) hipLaunchKernelGGL((maxpool2_kernel<unsigned short>), grid1(items), dim3(256), 0, s, (const unsigned short*)x, (unsigned short*)y, N, H, W, C, relu);
    else { mage_set_error("mage_maxpool2: bad dtype %d", dtype); return MAGE_EINVAL; }
    MAGE_CHECK_LAUNCH("mage_maxpool2");
    return MAGE_OK;
}

extern "C" int mage_upsample2(const void* x, void* y, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    MAGE_CHECK_ARG(x && y && N > 0 && C % 4 == 0, "mage_upsample2: bad arguments");
    const long items = (long)N * (H * 2) * (W * 2) * (C / 4);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MAGE_F32) hipLaunchKernelGGL((upsample2_kernel<float>), grid1(items), dim3(256), 0, s, (const float*)x, (float*)y, N, H, W, C);
    else if (dtype == MAGE_BF16) hipLaunchKernelGGL((upsample2_kernel<unsigned short>), grid1(items), dim3(256), 0, s, (const unsigned short*)x, (unsigned short*)y, N, H, W, C);
    else { mage_set_error("mage_upsample2: bad dtype %d", dtype); return MAGE_EINVAL; }
    MAGE_CHECK_LAUNCH("mage_upsample2");
    return MAGE_OK;
}

extern "C" int mage_relu(const void* x, void* y, int32_t dtype, int64_t n, void* stream) {
    MAGE_CHECK_ARG(x && y && n > 0 && n % 4 == 0, "mage_relu: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MAGE_F32) hipLaunchKernelGGL((map_kernel<float, float, true>), grid1(n / 4), dim3(256), 0, s, (const float*)x, (float*)y, (long)n);
    else if (dtype == MAGE_BF16) hipLaunchKernelGGL((map_kernel<unsigned short, unsigned short, true>), grid1(n / 4), dim3(256), 0, s, (const unsigned short*)x, (unsigned short*)y, (long)n);
    else { mage_set_error("mage_relu: bad dtype %d", dtype); return MAGE_EINVAL; }
    MAGE_CHECK_LAUNCH("mage_relu");
    return MAGE_OK;
}

extern "C" int mage_cast(const void* x, int32_t x_dtype, void* y, int32_t y_dtype, int64_t n, void* stream) {
    MAGE_CHECK_ARG(x && y && n > 0 && n % 4 == 0, "mage_cast: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == MAGE_F32 && y_dtype == MAGE_BF16)
        hipLaunchKernelGGL((map_kernel<float, unsigned short, false>), grid1(n / 4), dim3(256), 0, s, (const float*)x, (unsigned short*)y, (long)n);
    else if (x_dtype == MAGE_BF16 && y_dtype == MAGE_F32)
        hipLaunchKernelGGL((map_kernel<unsigned short, float, false>), grid1(n / 4), dim3(256), 0, s, (const unsigned short*)x, (float*)y, (long)n);
    else if (x_dtype == MAGE_F32 && y_dtype == MAGE_F32)
        hipLaunchKernelGGL((map_kernel<float, float, false>), grid1(n / 4), dim3(256), 0, s, (const float*)x, (float*)y, (long)n);
    else if (x_dtype == MAGE_F32 && y_dtype == MAGE_F16)
        hipLaunchKernelGGL((map_kernel<float, f16_t, false>), grid1(n / 4), dim3(256), 0, s, (const float*)x, (f16_t*)y, (long)n);
    else if (x_dtype == MAGE_F16 && y_dtype == MAGE_F32)
        hipLaunchKernelGGL((map_kernel<f16_t, float, false>), grid1(n / 4), dim3(256), 0, s, (const f16_t*)x, (float*)y, (long)n);
    else { mage_set_error("mage_cast: unsupported %d -> %d", x_dtype, y_dtype); return MAGE_EINVAL; }
    MAGE_CHECK_LAUNCH("mage_cast");
    return MAGE_OK;
}
