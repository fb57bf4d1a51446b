/*
 * mage_hip.h -- C ABI of libmage_hip.so: hand-written HIP (gfx950 / CDNA4) kernels for
 * the MAGE autoregressive video-token generation path.
 *
 * The reference (Youncy-Hu/MAGE) is pure PyTorch: it has NO FFI today.  Each entry
 * point below replaces the torch-op arithmetic at the cited reference site; the only
 * caller is the host-side mirror in mage_amd/modules/ (same class names and forward()
 * signatures as modules/mage_model.py / modules/vqvae_model.py), through ctypes
 * (mage_amd/_lib.py).  See INTEGRATION.md for the binding a reference maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (a torch tensor kept alive by
 *     the Python caller for the duration of the asynchronous call);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     kernels are enqueued on it and the call returns without synchronising;
 *   - return value: 0 on success, negative MAGE_E* code otherwise; mage_last_error()
 *     gives a thread-local message.  The ctypes shim maps codes to RuntimeError/ValueError;
 *   - activations are channels-last: a [.., C] tensor is a row-major [rows, C] matrix;
 *   - dtype tags: MAGE_F32 = IEEE fp32 (GEMMs run on v_mfma_f32_16x16x4_f32: exact fp32
 *     fma chains -- the parity mode), MAGE_BF16 = bf16 storage + v_mfma_f32_16x16x32_bf16
 *     with fp32 accumulation (the performance mode).  The residual stream, LayerNorm
 *     statistics, softmax, logits and all VQ arithmetic are fp32 in both modes.
 */
#ifndef MAGE_HIP_H
#define MAGE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAGE_ABI_VERSION 8

/* MAGE_BF16X3 / MAGE_F16X3: SPLIT-PRECISION operands -- the fast parity mode.  A logical fp32 matrix [rows, C] (C % 64 == 0, base
 * 256-byte aligned) is stored as two 16-bit pieces per element, x ~ hi + lo, per row as 64-column slabs [hi(64) | lo(64)] (so a row
 * is 2C 16-bit elements and leading dimensions of split tensors are given in 16-bit elements, normally 2C):
 *     MAGE_BF16X3: hi = bf16(x), lo = bf16(x - hi)           8 + 8 significand bits (relative representation error 2^-18)
 *     MAGE_F16X3:  hi = f16(x),  lo = f16((x - hi) * 2^11)   11 + 11 bits (2^-22), |x| clamped to 65504
 * mage_gemm on such operands runs THREE bf16/f16 MFMA products per K slab with fp32 accumulation,
 *     A W^T ~ A_hi W_lo^T + A_lo W_hi^T (all K slabs; f16: the sum is then scaled by 2^-11)  +  A_hi W_hi^T (all K slabs),
 * i.e. 3/16 of the exact-fp32 MFMA cost for an fp32-class result (the dropped lo*lo term and the representation error are 2^-18
 * resp. 2^-22 of a product; fp32's own accumulation error over K terms is of the same order as the f16 form's). */
/* MAGE_F16: IEEE half storage + v_mfma_f32_16x16x32_f16 with fp32 accumulation -- the bf16 kernels' schedules, LDS images and epilogues with
 * the other 16-bit type (same MFMA rate, 11 significand bits instead of 8: operand rounding 2^-12 instead of 2^-9; range +-65504, no clamp).
 * Accepted where the decoder stack of the generation path needs it: mage_gemm (plain rows and the padded-taps row-table form; act none /
 * QuickGELU; the LayerNorm-folded and x + Linear(.) forms; no split-K, no training forms, no head_w), mage_attention, mage_layernorm,
 * mage_row_stats, mage_table_conv, mage_embedding, mage_cast. */
enum { MAGE_F32 = 0, MAGE_BF16 = 1, MAGE_BF16X3 = 2, MAGE_F16X3 = 3, MAGE_F16 = 4 };
enum { MAGE_OK = 0, MAGE_EINVAL = -1, MAGE_EHIP = -2, MAGE_EUNSUPPORTED = -3 };
enum { MAGE_ACT_NONE = 0, MAGE_ACT_RELU = 1, MAGE_ACT_QUICKGELU = 2, MAGE_ACT_GELU_ERF = 3, MAGE_ACT_TANH = 4,
       MAGE_ACT_QUICKGELU_GRAD = 5 /* mage_gemm only: y = acc * QuickGELU'(y2), y2 = the saved pre-activation rows (bf16, READ): the
                                      data gradient of Linear -> QuickGELU -> Linear without an activation-backward pass */ };

int mage_abi_version(void);
const char* mage_last_error(void);
/* One-time per-device setup (allocates the 4 KiB zero page the gather loads use for padding and the deferred-error word).
 * Must be called once per process+device before any other call, outside stream capture.  Leaves the caller's current
 * device unchanged; returns MAGE_EUNSUPPORTED (every time) on a device that is not gfx950. */
int mage_init(int device);
/* Deferred device-side argument errors.  Kernels cannot return a code, and the host cannot see an index that lives in device
 * memory: mage_embedding (an id outside [0, n_table): where the reference's nn.Embedding raises IndexError) and
 * mage_cross_entropy (a target outside [0, K)) record the first such event in a per-device word and carry on with a clamped,
 * memory-safe value.  This call synchronises `stream`, returns MAGE_EINVAL with the message (and clears the word) if one
 * was recorded since the last check, MAGE_OK otherwise.  The host-side mirror calls it at the end of every public entry
 * (MAGE.forward, MAGE.autoregressive_generate, VectorQuantizedVAE.decode / forward). */
int mage_check_device_errors(void* stream);

/* Kernel-selection options (tuning, A/B tests, bisecting a regression): ONE table inside the library, filled once from the environment
 * variables MAGE_<NAME IN CAPITALS> the first time any entry point needs it, changed afterwards only through mage_set_option (no dispatch
 * function reads the environment; a change applies to every later call of the process).  Names:
 *   gemm_no_4w, gemm_no_4h, gemm_4h_plain, gemm4_train_forms, gemm_no_8phase, gemm_no_taps8, gemm_no_narrow, gemm_no_narrow_few, gemm_no_small, gemm_small_m,
 *   gemm_stagger_groups / _percent / _forced (MAGE_GEMM_STAGGER="G,percent"), gemm4_stagger_groups / _percent
 *   (MAGE_GEMM4_STAGGER), attn_no_mfma, attn_no_fewq, vq_no_mfma          -- what each one does: struct MageOptions in csrc/common.h and
 *   the table in INTEGRATION.md.  Unknown name: MAGE_EINVAL. */
int mage_set_option(const char* name, int32_t value);
int mage_get_option(const char* name, int32_t* value);

/* ---------------------------------------------------------------------------------------------
 * Fused GEMM / implicit-GEMM convolution on MFMA:   Y[yrow(m), n] = epi( sum_k A[arow(m,k), .] * W[n, k] )
 *
 * Replaces: every nn.Linear on the path (mage_model.py:22-26,332-335,348,375-376,385; the MHA
 * in/out projections inside nn.MultiheadAttention :20,:75; text encoder :193-207), the frame
 * conv3x3 (mage_model.py:485-488,587,648,675) and every Conv2d/ConvTranspose2d with C_in >= 8
 * of the VQ-VAE (vqvae_model.py:111-214), with BatchNorm(eval), bias, activation, positional
 * tables and the residual add fused into the epilogue.
 *
 * Kernels behind this entry point (csrc/gemm.hip, csrc/gemm4.hip; chosen from the descriptor, same bits per output element from all of them for
 * bf16 plain GEMMs): the lockstep persistent kernel (any dtype / gather / epilogue), its 8-phase ping-pong variant (bf16, >= 2 tiles of 256x256
 * per CU), the one-wave-per-SIMD variant (bf16, bias or LayerNorm-consuming epilogue, K in [256, 1024], >= 4 tiles per CU: the decoder's QKV and
 * c_fc; option gemm_no_4w disables it) and its split-half form with the epilogue under the K loop (csrc/gemm4h.hip: K = 512, 16-bit rows out;
 * by default the QuickGELU forms = c_fc; gemm_no_4h disables it, gemm_4h_plain sends QKV there too), the few-rows kernel (M <= 1024), the padded-taps forms and the split-precision forms.
 *
 * Row geometry.  A GEMM row m in [0, M) is decoded as img = m / (out_h*out_w),
 * oy = (m / out_w) % out_h, ox = m % out_w.  K = taps_h*taps_w*cin; k -> (ky, kx, ci), ci fastest:
 *     iy = oy*stride + dy0 + ky*dys,   ix = ox*stride + dx0 + kx*dxs      (zero outside [0,in_h)x[0,in_w))
 *     arow = img*a_img_stride + iy*in_w + ix + a_off
 *     yrow = img*y_img_stride + oy*y_mul_y + ox*y_mul_x + y_off
 * A plain Linear is taps_h=taps_w=1, out_h=in_h=1, out_w=in_w=P: rows are then regrouped P at a
 * time (x[:, 1:] views, writing frames into slots 1.. of the decoder input) without copies.
 *
 * Epilogue, in this order:  v = acc + bias[n];  v = v*scale[n] + shift[n] (BatchNorm eval);
 * v = act(v);  v += rowadd[((yrow / rowadd_div) % rowadd_mod), n];  v += residual[yrow, n];
 * v = relu(v) if post_relu;  store as y_dtype.  Null pointers skip a stage.  Y may alias residual.
 *
 * Padded-taps form (bf16): when the input is ZERO-PADDED (in_h = out_h + taps_h - 1, in_w = out_w + taps_w - 1, stride 1, dy0 = dx0 = 0,
 * cin % 64 == 0, M and N multiples of 256, packed output rows) every tap is a valid row and the convolution runs on the 8-phase
 * ping-pong kernel with one scalar offset per K slab; epilogue y = act(acc + bias) (act none / ReLU) or y = rowadd[..] + acc
 * (rowadd set, no bias / residual: the table is loaded into the accumulators before the K loop).  Anything else takes the generic
 * gather kernel.
 *
 * Split-precision form (dtype MAGE_BF16X3 / MAGE_F16X3): A and W are split tensors (see the dtype enum; lda / ldw in 16-bit elements,
 * ldw 0 = 2K), K and cin multiples of 64, plain rows or the padded-taps form only (with rowadd: M, N multiples of 256), epilogue
 * y = act(acc + bias) (act none / QuickGELU) or y = residual + acc + bias (fp32 residual) or y = rowadd[..] + acc; y_dtype MAGE_F32,
 * or the same split kind (N % 64 == 0, ldy in 16-bit elements, Y 256-byte aligned): the epilogue then writes the pieces of its fp32
 * result, i.e. the next GEMM's A operand, directly.
 *
 * Requirements: lda and cin multiples of 8 (bf16) / 4 (f32); N multiple of 8; ldy/ldr multiples of 4 (fp32) / 8 (bf16);
 * A, W, Y 16-byte aligned; W is [N][K] row-major in `dtype`.
 * ------------------------------------------------------------------------------------------- */
typedef struct mage_gemm_desc {
    int32_t dtype;                     /* MAGE_F32 | MAGE_BF16 | MAGE_BF16X3 | MAGE_F16X3: type of A and W and of the MFMA */
    int32_t M, N, K;
    const void* A;
    const void* W;
    void* Y;
    int32_t lda, ldy;                  /* in elements */
    int32_t y_dtype;                   /* MAGE_F32 | MAGE_BF16 (| the split kind of `dtype`) */
    int32_t out_h, out_w, in_h, in_w;
    int32_t a_img_stride, a_off;
    int32_t taps_h, taps_w, cin, stride;
    int32_t dy0, dx0, dys, dxs;
    int32_t y_img_stride, y_mul_y, y_mul_x, y_off;
    const float* bias;
    const float* scale;
    const float* shift;
    int32_t act;
    const float* rowadd;
    int32_t rowadd_div, rowadd_mod;
    const void* residual;
    int32_t ldr, res_dtype;
    int32_t post_relu;
    int32_t ldw;                       /* row stride of W in elements; 0 = K (W packed [N][K]) */
    /* Split-K (the weight-gradient GEMMs dW = dY^T X of the training path, whose contraction runs over all M tokens while the
     * output is only [N_out, K_out]): n_split > 1 computes n_split independent products in ONE launch,
     *     Y + s*y_split_stride  =  (A + s*a_split_stride) (*) (W + s*w_split_stride)^T      s = 0 .. n_split-1
     * (strides in elements; A and W then have lda / ldw > K and each slice contracts over K of their columns); the caller sums
     * the n_split partial outputs (mage_sum_partials) in a fixed order.  Plain form only: no gather, no epilogue extras. */
    int32_t n_split;                   /* 0 or 1 = no split */
    int64_t a_split_stride, w_split_stride, y_split_stride;
    /* LayerNorm folded around the decoder's GEMMs (bf16, plain rows, M and N multiples of 256; mage_model.py:35-53: the
     * x + dropout(attn(ln_1(x))) / x + mlp(ln_2(x)) chain).  Producer -- the x + Linear(.) GEMM that writes the fp32 stream: with y2
     * set it also writes a bf16 copy of the new rows (ldy2 elements per row) and ln_part[N/64][ln_part_rows][2] = (sum, sum of squares) of each
     * 64-column slice; mage_ln_stats reduces those to ln_stats[row][2] = (mean, rstd).  Consumer -- the Linear that follows the norm:
     * A = that bf16 copy, W = gamma * W (per input channel), bias = W beta + b, ln_colsum[n] = sum_k W'[n, k]; with ln_stats set the
     * epilogue computes rstd_m (acc - mean_m ln_colsum[n]) + bias[n] before the activation: LN(x) W^T + b without the LayerNorm pass.
     * bf16 stream: with ln_part set, y2 null and y_dtype MAGE_BF16 the producer writes the bf16 rows as its only output (the residual
     * may itself be that bf16 stream, res_dtype MAGE_BF16: x stays in bf16 between the blocks; a bf16 residual is also accepted by the
     * plain x + Linear(.) form without ln_part).  The partial sums are always those of the fp32 values before rounding. */
    void* y2;
    int32_t ldy2;
    int32_t res_half;                  /* general epilogue, out_h > 1: the residual lives at HALF resolution ([img, out_h/2, out_w/2, N] rows):
                                        * output pixel (oy, ox) adds residual pixel (oy/2, ox/2) -- nn.Upsample(scale_factor=2) of the
                                        * skip path folded into the convolution that consumes it (vqvae_model.py:147-166,203-209) */
    float* ln_part;
    const float* ln_stats;
    const float* ln_colsum;
    int32_t a_half;                    /* gather form: the input lives at HALF resolution ([img, in_h/2, in_w/2, cin] rows, a_img_stride = that
                                        * plane): tap pixel (iy, ix) of the in_h x in_w grid reads row (iy/2)*(in_w/2) + ix/2 -- nn.Upsample
                                        * (scale_factor=2, nearest) in front of a convolution folded into its gather (vqvae_model.py:203-209) */
    float ln_eps;                      /* consumer with ln_stats null and ln_part set: the epilogue takes (mean, rstd) of its rows straight from
                                        * the producer's partial sums (mage_ln_stats' arithmetic, eps = ln_eps; K / 64 slices per row) -- no
                                        * mage_ln_stats launch in between.  Few-rows GEMMs only: ask mage_gemm_is_small first.  (Built and measured
                                        * for the tiled kernels at the incremental loop's 16 k rows, round 5: the in-tile reduction costs the
                                        * consumers +5 .. +11 us per launch against the 7 us launch it removes -- not kept.) */
    const void* head_w;                /* bf16 padded-taps form with N == 256 (one column tile holds whole rows), bias, ReLU: a second, narrow Linear
                                        * taken on the tile before it leaves the CU.  head_w is bf16 [16][N]; the rows y = relu(acc + bias), rounded
                                        * to bf16 as a store would round them, are NOT written; Y (y_dtype MAGE_F32, ldy >= 16 floats) receives
                                        *     Y[yrow][t] = sum_n y[n] * head_w[t][n],  t = 0..15      (fp32 sums, fixed order)
                                        * (ldy == 4 with head_phases 0: only t = 0..3 are written, 16 bytes per row -- a head with <= 4 outputs)
                                        * -- the 4 x 4 taps of the VQ-VAE's last ConvTranspose2d (vqvae_model.py:187) computed inside the
                                        * sub-pixel GEMMs of the one before it (:184), whose 4x-resolution activation is then never stored.
                                        * With `residual` (bf16 rows [.., N]; res_half allowed; head_phases 0) the rows are
                                        *     y = relu((acc + bias) + residual)
                                        * -- the last DecoderBlock's closing 3x3 convolution + identity path + the decoder's ReLU with the 1x1 RGB
                                        * head taken on the tile (f8 VQ-VAE, vqvae_model.py:147-166,209-213): the 128 x 128 x 256 activation
                                        * (8 MB per frame) is never stored */
    int32_t head_phases;               /* with head_w: 0, or 4 = the four sub-pixel phases of that ConvTranspose2d(., ., 4, 2, 1) in ONE launch:
                                        * N = 4 * 256, W and bias hold the phases' [256][K] / [256] blocks in the order (py, px) = (0,0) (0,1)
                                        * (1,0) (1,1); column tile p reads its 2 x 2 window at a_off + py*in_w + px and writes the rows
                                        * y_off + py*(y_mul_y/2) + px*(y_mul_x/2) (y_mul_x == 2): four launches' tiles, bit for bit, from one tile
                                        * list in which a frame's four phases are neighbours (its padded rows are fetched once) */
    int32_t a_relu;                    /* 16-bit plain GEMM with N <= 128 (the 256 x 64 tile): the product is taken over relu(A) -- the ReLU in front of a
                                        * bottleneck block's first 1x1 convolution (vqvae_model.py:147-166: Sequential(ReLU, Conv2d 1x1, ...)) applied to
                                        * the operand fragments, so that relu(x) is never stored next to x (which the identity path reads).  Refused
                                        * on every other form */
    int64_t ln_part_rows;              /* rows of the ln_part buffer = the stride (in float2) between two 64-column slices: slice-major, so that the 16
                                        * rows of an accumulator tile leave as ONE 128-byte store (row-major partial sums left as 16 scattered 8-byte
                                        * pieces per store) and mage_ln_stats reads them coalesced.  Required (> the largest output row) with ln_part */
} mage_gemm_desc;

int mage_gemm(const mage_gemm_desc* desc, void* stream);
/* 1 if a bf16 plain GEMM of this size (lean epilogue: bias / x + Linear(.) / the LayerNorm-folded forms) runs on the few-rows kernel
 * on the current device (one clip per call), 0 if on the tiled kernels, < 0 on error */
int mage_gemm_is_small(int32_t M, int32_t N, int32_t K);
/* stats[row] = (mean, rstd) from the producer GEMM's partial sums part[n_slices][rows][2] (slice-major: mage_gemm_desc::ln_part_rows = rows):
 * mean = sum_s part[s][row][0] / C, var = sum_s part[s][row][1] / C - mean^2, rstd = 1 / sqrt(max(var, 0) + eps); fixed order (s ascending). */
int mage_ln_stats(const float* part, int64_t rows, int32_t n_slices, int32_t C, float eps, float* stats, void* stream);
/* stats[row] = (mean, rstd) of bf16 rows x[row][0..C) (fp32 sums): the LayerNorm in front of the decoder's first Linear when the residual
 * stream starts in bf16 (mage_model.py:35-36,47 on the rows mage_model.py:375-378 assembles); consumed like mage_ln_stats' output */
int mage_row_stats(const void* x, int32_t dtype, int64_t rows, int32_t C, int64_t ldx, float eps, float* stats, void* stream);

/* y = the split-precision form (kind MAGE_BF16X3 | MAGE_F16X3) of fp32 rows x [rows, C] (row stride ldx floats; C % 64 == 0);
 * y rows have ldy 16-bit elements (>= 2C), y 256-byte aligned.  Weights are split once per state_dict (derived caches); activations
 * are normally written split by their producers (mage_layernorm, mage_attention, mage_embedding, the GEMM epilogue). */
int mage_split(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t C, int32_t kind, void* stream);

/* mage_split with a row map and an optional ReLU: y row  (i / group)*group_stride + ((i % group) / inner)*inner_stride + (i % group) % inner + off
 * (mage_embedding's map: the interior of a zero-padded frame buffer) <- split(relu ? max(x row i, 0) : x row i); x_relu (optional, may be x):
 * the same rows written back as fp32 -- a ResBlock's in-place ReLU (vqvae_model.py:111-124): its skip path and its 3x3 convolution's
 * padded split input in one pass. */
int mage_split_rows(const float* x, int64_t ldx, void* y, int64_t rows, int32_t C, int32_t kind, int32_t relu, int64_t group,
                    int64_t group_stride, int64_t off, int64_t inner, int64_t inner_stride, float* x_relu, void* stream);

/* LayerNorm over the last dim of fp32 rows; y may be fp32 (may alias x), bf16, or split (MAGE_BF16X3 / MAGE_F16X3: C % 64 == 0).
 * Replaces nn.LayerNorm at mage_model.py:21,27,84,204,206 and inside nn.TransformerEncoderLayer. */
int mage_layernorm(const float* x, const float* gamma, const float* beta, void* y, int32_t y_dtype,
                   int64_t rows, int32_t C, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-head attention core softmax(q k^T * scale + mask) v for SHORT key sets (nk <= 64),
 * head_dim = 32, fp32 arithmetic, no permute copies: sequences are strided row sets.
 * Replaces the scaled-dot-product inside nn.MultiheadAttention for the axial blocks
 * (mage_model.py:31-33,47-52: axis L with the causal mask :367-372, axes H and W), the text
 * encoder self-attention with key padding (:237-244) and the motion-anchor cross-attention (:87-92).
 *
 * Sequence s in [0, n_seq): outer = s / inner, in = s % inner.
 *   query i  -> row  outer*q_outer_stride  + in + i*q_axis_stride     of q  (and of out)
 *   key   j  -> row  outer*kv_outer_stride + in + j*kv_axis_stride    of k, v
 * Head h uses columns [h*32, h*32+32).  causal: key j visible to query i iff j <= i + (nk - nq) (the mask is aligned to
 * the LAST key, so nq == nk is the usual lower triangle and nq < nk is a query block appended to a key cache).
 * kv_len (optional, int32 [ceil(n_seq / kv_len_div)]): only keys j < kv_len[s / kv_len_div] are visible.
 * ------------------------------------------------------------------------------------------- */
typedef struct mage_attn_desc {
    int32_t dtype;                     /* element type of q, k, v and out; MAGE_F16X3: q, k, v are SPLIT rows (ld* in 16-bit elements; head h's
                                        * columns through the slab map) and out_split must be MAGE_F16X3: the fast parity mode's axial attention
                                        * on the matrix cores (three f16 MFMA passes per product, nq, nk <= 32) */
    const void* q;
    const void* k;
    const void* v;
    void* out;
    int32_t ldq, ldk, ldv, ldo;
    int32_t n_seq, inner;
    int32_t nq, nk, n_head;
    int32_t q_outer_stride, q_axis_stride;
    int32_t kv_outer_stride, kv_axis_stride;
    int32_t causal;
    const int32_t* kv_len;
    int32_t kv_len_div;
    float scale;
    int32_t out_split;                 /* 0: out has `dtype`; MAGE_BF16X3 / MAGE_F16X3 (dtype MAGE_F32 only): out is written as split
                                        * rows (ldo in 16-bit elements = 2 * logical width, 256-byte aligned): out_proj's A operand */
    /* Dropout on the attention PROBABILITIES (nn.MultiheadAttention(dropout=p) in train(): the text encoder's
     * nn.TransformerEncoderLayer, mage_model.py:193-199): out = (softmax(.) * keep / (1 - p)) v with the stateless mask
     * keep(s, h, i, j) = hash(drop_seed, ((s * n_head + h) * nq + i) * nk + j) >= p * 2^32, recomputed by mage_attention_bwd.
     * fp32 thread-per-query kernels only (drop_p = 0: off). */
    float drop_p;
    uint64_t drop_seed;
    /* Row map of `out` when it differs from q's (both 0: out row of query i = q's row): query i of sequence s -> row
     * outer*o_outer_stride + in + i*o_axis_stride.  The incremental loop's temporal blocks read q from the [q | k | v] cache slots
     * (q_outer_stride = L*hw) and write the packed rows of the new positions (o_outer_stride = P*hw). */
    int32_t o_outer_stride, o_axis_stride;
} mage_attn_desc;

int mage_attention(const mage_attn_desc* desc, void* stream);

/* out[orow(i), :] = act(table[ids[i], :]),  orow(i) = (i / group)*group_stride + ((i % group) / inner)*inner_stride + (i % group) % inner + off
 * (inner <= 0: one level, orow = (i / group)*group_stride + i % group + off).  Two levels write an image's h x w tokens into the
 * interior of a zero-padded (h+2) x (w+2) frame buffer (group = h*w, group_stride = (h+2)(w+2), inner = w, inner_stride = w+2,
 * off = w+3): the input layout of the padded-taps convolution of mage_gemm.
 * Replaces nn.Embedding lookups: visual_token_embedding (mage_model.py:581,644,682), the codebook
 * gather of VectorQuantizedVAE.decode (vqvae_model.py:240; relu=1 folds the in-place ReLU that
 * opens the first decoder ResBlock :113), text token embedding (:226).  An id outside [0, n_table) is read as the nearest valid
 * row AND recorded for mage_check_device_errors (the reference raises IndexError). */
int mage_embedding(const int64_t* ids, const float* table, void* out, int32_t out_dtype, int64_t n,
                   int32_t C, int32_t n_table, int32_t relu, int64_t group, int64_t group_stride, int64_t off,
                   int64_t inner, int64_t inner_stride, void* stream);

/* k x k convolution (stride 1, zero padding (k-1)/2, no bias) of nn.Embedding rows as a TABLE SUM -- the frame convolution
 * `conv(visual_token_embedding(ids))` of mage_model.py:581,586-588,674-676 (and, folded in by linearity, the `in_linear` that consumes it
 * :375-376 and its bias): the input has only n_codes distinct vectors, so
 *     y[yrow(m), :] = act(pos[p, :] + bias[:] + sum_{taps inside the image} table[tap][ids[img, neighbour], :] + rowadd[(yrow / rowadd_div) % rowadd_mod, :])
 * m = img*H*W + p, yrow(m) = (m / group)*y_group_stride + m % group + y_off; table [taps_h*taps_w][n_codes][C] (fp32 or bf16) = W_tap emb[code]
 * precomputed once per weights by the caller (derived cache; 9.4 MB at the MNIST config: resident in L2 / Infinity Cache), pos [H*W][C], bias [C]
 * and rowadd optional fp32 tables; act = ReLU if relu; fp32 sums in a fixed order; y fp32 or bf16.  0 FLOP on the matrix cores instead of
 * 2*9*C*C per pixel.  Also the first 3x3 convolution of the f4 VQ-VAE decoder (vqvae_model.py:111-124,180: conv(relu(codebook[ids]))).
 * An id outside [0, n_codes) is recorded for mage_check_device_errors like mage_embedding's. */
int mage_table_conv(const int64_t* ids, int64_t n_img, int32_t H, int32_t W, int32_t taps_h, int32_t taps_w, const void* table,
                    int32_t table_dtype, int32_t n_codes, int32_t C, const float* pos, const float* bias, int32_t relu, const float* rowadd,
                    int64_t rowadd_div, int32_t rowadd_mod, void* y, int32_t y_dtype, int64_t ldy, int64_t group, int64_t y_group_stride,
                    int64_t y_off, void* stream);

/* The FIRST ResBlock of the f4 VQ-VAE decoder in one launch (vqvae_model.py:111-124 applied at :180 to codebook.embedding(latents); bf16):
 * its input is x = relu(codebook[ids]) (the block's in-place ReLU), so with table[tap][code] = (BatchNorm-folded) W3_tap relu(codebook[code])
 *     t = relu(bias3 + sum_{taps inside the image} table[tap][ids[neighbour]])        rounded to bf16 (mage_table_conv's sum, same order)
 *     y = max(x + ((t W1^T + b1) * scale1 + shift1), post_relu ? 0 : -inf)            x rounded to bf16 first, as a bf16 frame buffer holds it
 * y row of pixel (img, py, px) = img*y_img_stride + py*y_row_pitch + px + y_off (ldy bf16 elements per row): the interior of a zero-padded
 * frame buffer.  Replaces mage_embedding + mage_table_conv + the 1x1 mage_gemm (general epilogue) with the same bits out: t and x never
 * reach HBM.  Built for W == 16, even H, C == 256; table bf16 [9][n_codes][C], codebook fp32 [n_codes][C], w1 bf16 [C][C] row-major;
 * scale1 / shift1 optional (both or none).  An id outside [0, n_codes) is recorded for mage_check_device_errors like mage_embedding's. */
int mage_resblock_table(const int64_t* ids, int64_t n_img, int32_t H, int32_t W, const void* table, int32_t n_codes, int32_t C,
                        const float* bias3, const float* codebook, const void* w1, const float* b1, const float* scale1, const float* shift1,
                        int32_t post_relu, void* y, int64_t ldy, int64_t y_img_stride, int64_t y_row_pitch, int64_t y_off, void* stream);

/* The tail of a ResBlock whose 3x3 convolution ran as a GEMM (vqvae_model.py:111-124; bf16): t = relu(BN(conv3x3(relu x))) in plain rows
 * [n_img*H*W][C] (lda), x and y in frame buffers whose row of pixel (img, py, px) is img*img_stride + py*row_pitch + px + off (ldr / ldy):
 *     y = max(x + ((t W1^T + b1) * scale1 + shift1), post_relu ? 0 : -inf)
 * -- the 1x1 mage_gemm with scale / shift, bf16 residual and post_relu, same bits, as an HBM-bound row kernel (whole 512-byte rows in and
 * out, W1 in registers).  C == 256, n_img*H*W a multiple of 64; w1 bf16 [C][C]; scale1 / shift1 optional; y may alias residual. */
int mage_resblock_rows(const void* t, int64_t lda, const void* w1, const float* b1, const float* scale1, const float* shift1,
                       const void* residual, int64_t ldr, int32_t post_relu, void* y, int64_t ldy, int64_t n_img, int32_t H, int32_t W,
                       int32_t C, int64_t img_stride, int64_t row_pitch, int64_t off, void* stream);

/* Nearest codebook entry, reference formula and tie-break (vqvae_model.py:8-25):
 *   dist[m,k] = (|c_k|^2 + |z_m|^2) - 2 * <z_m, c_k>,  idx[m] = first k attaining the minimum.
 * z [M, D] fp32 rows (channels-last encoder output), codebook_t [D, K] fp32 (transposed copy),
 * c2 [K] = row sums of squares.  margin (optional, [M] fp32) receives second-best minus best. */
int mage_vq_nearest(const float* z, const float* codebook_t, const float* c2, int64_t M, int32_t D, int32_t K,
                    int64_t* idx, float* margin, void* stream);
/* c2[k] = sum_d codebook[k, d]^2 and codebook_t = codebook^T (derived caches of the codebook). */
int mage_vq_prepare(const float* codebook, int32_t K, int32_t D, float* codebook_t, float* c2, void* stream);

/* Row argmax with first-maximum tie-break (torch.max(prediction, -1)[1], mage_model.py:681,687).
 * Row i in [0, rows) is read from logits row  (i / group)*in_group_stride + i % group + in_off  (fp32, leading
 * dim ld) and its index written to out[(i / group)*out_group_stride + i % group + out_off] (int64): this is how
 * frame i of every clip is picked out of [B, L-1, hw, K] logits and written into slot i+1 of the token buffer
 * without copies.  margin (optional, [rows] fp32) receives best minus second-best. */
int mage_argmax(const float* logits, int64_t rows, int32_t K, int64_t ld, int64_t group, int64_t in_group_stride,
                int64_t in_off, int64_t* out, int64_t out_group_stride, int64_t out_off, float* margin, void* stream);

/* Mean cross entropy over rows (F.cross_entropy, mage_model.py:618): row_loss[i] = logsumexp(logits[i]) -
 * logits[i, target[i]] (workspace, [rows] fp32), loss_mean[0] = mean_i row_loss[i] (fixed-order fp64 sum:
 * deterministic).  A target outside [0, K) is recorded for mage_check_device_errors. */
int mage_cross_entropy(const float* logits, const int64_t* target, int64_t rows, int32_t K, float* row_loss,
                       float* loss_mean, void* stream);

/* ------------------------------------------------------------------------------------- direct convs
 * Small-channel ends of the VQ-VAE that are HBM-bound, not GEMMs.
 * mage_conv_in:  NCHW fp32 image [N, cin<=4, H, W] -> channels-last [N, OH, OW, cout] (y_dtype),
 *   y = act((conv(x) + bias) * scale + shift).  f4 stem Conv2d(1,dim,4,2,1)+BN+ReLU (vqvae_model.py:173-175),
 *   f8 stem Conv2d(3,dim,7,padding=3) (:193).  weight_t is the transposed copy [cin, kh, kw, cout] fp32
 *   (lanes = output channels read it coalesced).  y_dtype may be a split kind (MAGE_F16X3 / MAGE_BF16X3).  s2d = 1: the output is written
 *   as OFFSET SPACE-TO-DEPTH rows [N, OH/2+1, OW/2+1, 4*cout] (block (R, C) = pixels (2R-1..2R, 2C-1..2C); the caller zeroes the buffer
 *   once: border sub-blocks are never written), so that a following Conv2d(., ., 4, 2, 1) (vqvae_model.py:176) is a 2x2 / stride-1
 *   window over blocks = the padded-taps form of mage_gemm with cin = 4*cout.
 * mage_conv_out: channels-last [N, IH, IW, cin] (x_dtype) -> NCHW fp32 [N, cout<=4, OH, OW], y = tanh(.):
 *   transposed=1: ConvTranspose2d(dim, cout, 4, 2, 1) (vqvae_model.py:187-188), weight_t [4, 4, cout, cin]
 *   (= torch weight [cin, cout, ky, kx] permuted to ky, kx, cout, cin);
 *   transposed=0: Conv2d(dim, cout, 1) (:212-213), weight_t [cout, cin]. */
int mage_conv_in(const float* x, const float* weight_t, const float* bias, const float* scale, const float* shift,
                 void* y, int32_t y_dtype, int32_t N, int32_t cin, int32_t H, int32_t W, int32_t cout,
                 int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t act, int32_t s2d, void* stream);
int mage_conv_out(const void* x, int32_t x_dtype, const float* weight_t, const float* bias, float* y,
                  int32_t N, int32_t IH, int32_t IW, int32_t cin, int32_t cout, int32_t transposed, void* stream);
/* The same ConvTranspose2d(dim, cout, 4, 2, 1) + Tanh head (vqvae_model.py:187-189) as GEMM + fold, which reads the
 * wide activation exactly once: taps = mage_gemm(x [N*IH*IW, cin], weight_t viewed as [16*cout, cin]) holds, per INPUT
 * pixel, its product with each of the 16 kernel taps (column (ky*4 + kx)*cout + co, fp32); this call sums the (up to) 4
 * taps that land on every OUTPUT pixel (oy = 2*iy - 1 + ky), adds the bias, applies tanh and writes NCHW fp32
 * [N, cout<=4, 2*IH, 2*IW]. */
int mage_convt_fold_tanh(const float* taps, const float* bias, float* y, int32_t N, int32_t IH, int32_t IW,
                         int32_t cout, void* stream);

/* Channels-last elementwise helpers of the f8 stack (vqvae_model.py:194-210): 2x2 max-pool,
 * nearest 2x upsample, and a ReLU'd copy (the non-in-place ReLU that opens every Encoder/DecoderBlock). */
int mage_maxpool2(const void* x, void* y, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t relu,
                  void* stream);
int mage_upsample2(const void* x, void* y, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int mage_relu(const void* x, void* y, int32_t dtype, int64_t n, void* stream);
/* y = x converted between fp32 and bf16 (n elements). */
int mage_cast(const void* x, int32_t x_dtype, void* y, int32_t y_dtype, int64_t n, void* stream);

/* ADAIN2D (mage_model.py:299-314): out[b,p,c] = gamma[b,p,c] * (x[b,p,c]-mean[b,c])*rstd[b,c] + beta[b,p,c]
 * with InstanceNorm2d statistics over the P = H*W positions (biased variance, eps 1e-5). fp32, channels-last. */
int mage_adain(const float* x, const float* gamma, const float* beta, float* out, int32_t B, int32_t P, int32_t C,
               float eps, void* stream);

/* x[b, p, :] += s[b] * vec[:]   (speed embedding, mage_model.py:666-668). fp32. */
int mage_add_scaled_rowvec(float* x, const float* s, const float* vec, int32_t B, int32_t P, int32_t C, void* stream);

/* x[r, :] = x[r, :] * rs[r] + table[(r / div) % mod, :]   (rs and/or table may be null), fp32, in place.
 * Text encoder: + positions (mage_model.py:227-228) and the padding-row zeroing (:233-235). */
int mage_row_affine(float* x, const float* rs, const float* table, int64_t rows, int32_t C, int32_t div, int32_t mod,
                    void* stream);

/* Caption bookkeeping of the text encoder (mage_model.py:233-239) in one launch: keep[b*S + s] = (ids[b][s] != padding_idx) as 1.0 / 0.0 (the row
 * scale of mage_row_affine that zeroes padded rows), kv_len[b] = number of kept tokens (mage_attention's kv_len).  Either output may be null. */
int mage_caption_mask(const int64_t* ids, int32_t B, int32_t S, int64_t padding_idx, int32_t* kv_len, float* keep, void* stream);
/* Strided device-to-device block copy on the stream (hipMemcpy2DAsync): `height` rows of `width_bytes` bytes, pitches in bytes.  Assembles the
 * [B, L, C, H, W] result of autoregressive_generate (mage_model.py:691: the passed-through first frame + the decoded frames) without a compute
 * kernel. */
int mage_copy2d(void* dst, int64_t dst_pitch, const void* src, int64_t src_pitch, int64_t width_bytes, int64_t height, void* stream);

/* MAGE+ head (mage_model.py:350-354,387-388): y = SiLU(GroupNorm(groups, C)(x)) with the statistics of each sample taken
 * over rows_per_sample rows x (C/groups) channels.  x fp32 rows; sample b uses rows [b*sample_stride_rows + row_off, +rows_per_sample)
 * (this is how x[:, 1:] of the [B, L, hw, C] decoder stream is addressed without a copy); y is packed [n_samples*rows_per_sample, C]
 * in y_dtype; stats is a [n_samples, groups, 2] fp32 workspace (mean, rstd).  Two-pass, fixed-order reductions. */
int mage_groupnorm_silu(const float* x, int64_t sample_stride_rows, int64_t row_off, int32_t n_samples, int32_t rows_per_sample,
                        int32_t C, int32_t groups, const float* gamma, const float* beta, float eps, float* stats, void* y,
                        int32_t y_dtype, void* stream);

/* GroupNorm of the Conv3d video prior (BasicBlock, mage_model.py:264-297: GroupNorm(16, C) after each Conv3d, then ReLU, or the
 * sum with the downsample branch then ReLU): same statistics and addressing of x as mage_groupnorm_silu, plus
 *   y[b*y_sample_stride_rows + y_row_off + r, :] = act(GroupNorm(x)[b, r, :] + residual[b*rows_per_sample + r, :])
 * residual optional (fp32, packed); act 0 none / 1 ReLU / 2 SiLU.  The output addressing lets a block write straight into the
 * zero-padded frame buffer the next temporal convolution gathers from. */
int mage_groupnorm_act(const float* x, int64_t sample_stride_rows, int64_t row_off, int32_t n_samples, int32_t rows_per_sample,
                       int32_t C, int32_t groups, const float* gamma, const float* beta, float eps, float* stats,
                       const float* residual, int32_t act, void* y, int32_t y_dtype, int64_t y_sample_stride_rows,
                       int64_t y_row_off, void* stream);

/* Reparameterisation of MAGE.reparameterize (mage_model.py:569-573) and the summand of the KL term (:623), fp32:
 *   out = eps * exp(0.5 * logvar) + mu;   kl_sum[b] = sum over the n elements of sample b of (1 + logvar - mu^2 - exp(logvar))
 * mu, logvar, eps, out: [B, n] (any common layout); kl_sum [B]. */
int mage_reparam_kl(const float* mu, const float* logvar, const float* eps, float* out, float* kl_sum, int32_t B, int64_t n,
                    void* stream);

/* out[0] = mean over rows x cols of (a[r*lda + c] - b[r*ldb + c])^2, fp32 inputs, fp64 fixed-order accumulation
 * (F.mse_loss of the MAGE+ latent prediction, mage_model.py:620).  workspace: 256 doubles. */
int mage_mse(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int32_t cols, double* workspace, float* out,
             void* stream);


/* =============================================================================================================================
 * Training path: backward of MAGE.forward (main_mage.py:150-153 `loss.backward(); optimizer.step()`).
 * Dense gradients reuse mage_gemm: dX = dY W on a transposed weight copy; dW = dY^T X as transposes + ONE split-K launch
 * (mage_gemm_desc::n_split) + mage_sum_partials.  The entry points below are the rest of the backward pass.
 * =========================================================================================================================== */

/* y[(c + y_row0)*ldy + m] = x[arow(m)*ldx + c] for m < M, 0 for M <= m < Mp (zero tail up to the padded width the split-K GEMM
 * reads).  arow(m) is mage_gemm's implicit-GEMM row map for ONE tap: m -> (img, oy, ox) over an out_h x out_w plane,
 * arow = img*img_stride + (oy+dy)*in_w + (ox+dx) + a_off, zero outside [0,in_h) x [0,in_w).  Plain transpose: out_h = 1,
 * out_w = M, dy = dx = 0, stride = 1.  The conv3x3 weight gradient (mage_model.py:485-488) stacks nine calls (dy, dx = tap - 1,
 * y_row0 = tap*C); with `stride` the gathered pixel is (oy*stride+dy, ox*stride+dx): the 4x4 / stride-2 convolutions and
 * transposed convolutions of the VQ-VAE (vqvae_model.py:173-188).  dtype: MAGE_F32 | MAGE_BF16 (x and y). */
int mage_transpose(const void* x, int32_t dtype, int64_t ldx, void* y, int64_t ldy, int64_t y_row0, int64_t M, int64_t Mp, int32_t C,
                   int32_t out_h, int32_t out_w, int32_t in_h, int32_t in_w, int64_t img_stride, int64_t a_off, int32_t dy, int32_t dx,
                   int32_t stride, void* stream);
/* mage_transpose of bf16 rows (out_h = 1: row m -> (m / out_w) * img_stride + m % out_w + a_off, no tap shift) that also returns the
 * column sums of x: colsum[p][c] = sum over the p-th group of 16 consecutive 64-row tiles of x[., c] (n_part = ceil(ceil(Mp / 64) / 16)
 * partial rows for mage_sum_partials): the bias gradient db = sum_m dY[m, :] of a Linear in the same pass as the transposed copy of dY
 * that its weight gradient needs. */
int mage_transpose_colsum(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t M, int64_t Mp, int32_t C, int32_t out_w,
                          int64_t img_stride, int64_t a_off, float* colsum, int32_t n_part, void* stream);
/* Weight gradient WITHOUT transposed copies: partials[s][n][k] = sum over tokens t in [s*tokens_per_split, (s+1)*tokens_per_split) ∩ [0, T)
 * of dY[t*lda + n] * X[t*ldb + k]  (bf16 operands, row-major over the token index; fp32 accumulation and output; N and K multiples of
 * 256, tokens_per_split a multiple of 64, n_split * tokens_per_split >= T); the caller adds the n_split partials (mage_sum_partials).
 * dW = dY^T X of every nn.Linear of the decoder stack (loss.backward(), main_mage.py:152): tiles go to LDS as stored and the MFMA
 * fragments are read with the transposing LDS load (ds_read_b64_tr_b16).  db_partials (optional, [n_split][N]): the bias gradient's
 * column sums of dY per slice, added up from the fragments the kernel holds anyway.  mage_colsum: the same column sums on their own,
 * partials[p][c] over the p-th of n_part row chunks (fp32; finished by mage_sum_partials). */
int mage_gemm_tn(const void* dY, int64_t lda, const void* X, int64_t ldb, int64_t T, int32_t N, int32_t K, int32_t n_split,
                 int64_t tokens_per_split, float* partials, float* db_partials, void* stream);
int mage_colsum(const void* x, int64_t ld, int64_t T, int32_t C, float* partials, int32_t n_part, void* stream);
/* Row sums (fp32, fixed order): bias gradients db = column sums of dY, taken from the transposed dY.  The n columns are cut into
 * n_chunk chunks: out[chunk*rows + r] = sum over chunk of x[r*ld + c]; n_chunk > 1 is finished by mage_sum_partials. */
int mage_row_sum(const void* x, int32_t dtype, int64_t ld, int64_t n, int32_t rows, float* out, int32_t n_chunk, void* stream);
/* out[i] = (accumulate ? out[i] : 0) + sum_{s < n_part} part[s*stride + i]: the split-K partial products, the LayerNorm
 * gamma/beta partials.  Fixed order: deterministic. */
int mage_sum_partials(const float* part, int64_t stride, int32_t n_part, int64_t n, float* out, int32_t accumulate, void* stream);
/* nn.LayerNorm backward (statistics recomputed from the saved input x [rows, C] fp32): dx (+)= dLN/dx, and per-workgroup
 * partial sums partials[n_part][2][C] of (dgamma, dbeta) for mage_sum_partials.  dy in dy_dtype.
 * dx_bf16 (optional, [rows, C] bf16): the updated dx rows once more, through mage_dropout's mask (p, seed; p = 0: a plain cast) -- the
 * gradient stream as the operand of the next branch's GEMMs (x + dropout(Linear(.)), mage_model.py:48,52) without a separate pass. */
int mage_layernorm_bwd(const float* x, const float* gamma, const void* dy, int32_t dy_dtype, float* dx, float* partials, int32_t n_part,
                       int64_t rows, int32_t C, float eps, int32_t accumulate, void* dx_bf16, float p, uint64_t seed, void* stream);
/* y = act(x) and dx = dy * act'(x) elementwise (x = the saved pre-activation; QuickGELU mage_model.py:11-13, erf-GELU of the text
 * encoder, ReLU -- for which the post-activation output serves equally).  mage_act_bwd also takes MAGE_ACT_TANH, with x = the OUTPUT
 * y = tanh(.): dx = dy (1 - y^2).  n % 4 == 0. */
int mage_act(const void* x, void* y, int32_t dtype, int64_t n, int32_t act, void* stream);
int mage_act_bwd(const void* x, const void* dy, void* dx, int32_t dtype, int64_t n, int32_t act, void* stream);
/* F.cross_entropy backward (mage_model.py:618): dlogits = (softmax(logits) - onehot(target)) * grad_out[0] / rows, written in
 * dl_dtype (the A operand of the head's dX / dW GEMMs).  grad_out: device pointer to the upstream scalar gradient. */
int mage_cross_entropy_bwd(const float* logits, const int64_t* target, int64_t rows, int32_t K, const float* grad_out, void* dlogits,
                           int32_t dl_dtype, void* stream);
/* nn.Embedding backward: dtable[ids[i], :] += dout[orow(i), :] (orow as in mage_embedding; ids equal to padding_idx (< 0: none)
 * contribute nothing, as nn.Embedding(padding_idx=...) does).  fp32 atomics -- or, for tables of up to 512 rows (C % 64 == 0) with `scratch`
 * of at least min(64, ceil(n / 4096)) * n_table * C floats (16-byte aligned), DETERMINISTIC: per-chunk partial tables whose entries add
 * the chunk's rows in ascending order (no atomics between waves), summed in chunk order -- a fixed-order fp32 sum, bit-identical from run
 * to run (what makes the in-tree training reproducible). */
int mage_embedding_bwd(const int64_t* ids, const void* dout, int32_t dout_dtype, float* dtable, int64_t n, int32_t C, int32_t n_table,
                       int64_t padding_idx, int64_t group, int64_t group_stride, int64_t off, float* scratch, int64_t scratch_floats,
                       void* stream);
/* out[g, :] = sum over rows r with (r / div) % mod == g of w(r) x[r, :], w(r) = row_scale ? row_scale[r / row_scale_div] : 1.
 * Gradients of the broadcast row tables: T / H / W positional embeddings (mage_model.py:338,489-492), text positions, and (mod = 1,
 * row_scale = speed) the speed embedding (:666-668).  The rows of a group are cut into n_chunk chunks, out[chunk][g][c] holds the
 * partial sums (n_chunk > 1: finished by mage_sum_partials). */
int mage_group_rowsum(const void* x, int32_t dtype, int64_t rows, int32_t C, int64_t div, int64_t mod, const float* row_scale,
                      int64_t row_scale_div, float* out, int32_t n_chunk, void* stream);
/* Backward of mage_attention: same descriptor (q, k, v as in the forward call; desc->out is unused), dout addressed like the
 * forward's out (ldo), gradients written with the addressing of q (dq, ld_dq) and of k / v (dk, dv, ld_dk, ld_dv) in desc->dtype.
 * P is recomputed in fp32; fixed-order sums (deterministic). */
int mage_attention_bwd(const mage_attn_desc* desc, const void* dout, void* dq, void* dk, void* dv, int32_t ld_dq, int32_t ld_dk,
                       int32_t ld_dv, void* stream);
/* nn.Dropout in training mode as a stateless mask: y = (accumulate ? y : 0) + x * keep(seed, i) / (1 - p).  The backward pass
 * calls it again with the same seed (no mask tensor). */
int mage_dropout(const void* x, int32_t x_dtype, void* y, int32_t y_dtype, int64_t n, float p, uint64_t seed, int32_t accumulate,
                 void* stream);
/* y = r + dropout(x) with the same mask as mage_dropout(x, ., p, seed): the residual add x + dropout(Linear(.)) of a block
 * (mage_model.py:48,52) in one pass (r, y fp32; may not alias x).  y_bf16 (optional): the same rows once more as bf16 (the last block's
 * output is the head GEMM's operand). */
int mage_dropout_add(const void* x, int32_t x_dtype, const float* r, float* y, void* y_bf16, int64_t n, float p, uint64_t seed, void* stream);
/* The same residual add followed by the LayerNorm that opens the next branch, one pass: y = r + dropout(x) (fp32 rows [rows, C]) and
 * yn = LayerNorm(y; gamma, beta, eps) in yn_dtype (mage_model.py:48-52: x = x + attn(ln_1(x)); x = x + mlp(ln_2(x))).  x in x_dtype
 * (fp32, or bf16 with yn bf16); two-pass statistics as mage_layernorm. */
int mage_dropout_add_layernorm(const void* x, int32_t x_dtype, const float* r, float* y, const float* gamma, const float* beta, void* yn,
                               int32_t yn_dtype, int64_t rows, int32_t C, float eps, float p, uint64_t seed, void* stream);
/* BatchNorm2d in TRAINING mode (stage-1 VQ-VAE training, train_vqvae.py:13-35; vqvae_model.py:112-119,174,186) on channels-last
 * rows [rows, C] fp32: batch statistics are column reductions.  mage_bn_colreduce writes per-workgroup partial column sums
 * partials[n_part][NOUT][C] for mage_sum_partials (fixed order): mode 0: sum x; mode 1: sum (x - mean)^2 (two-pass variance);
 * mode 2 (backward): sum g and sum g*xhat with g = dy * (mask ? mask > 0 : 1) -- `mask` is the layer's post-ReLU output when a
 * ReLU follows the norm.  mage_bn_apply: y = [relu]((x - mean) rstd gamma + beta [+ residual]) (the ResBlock's skip, vqvae_model.py:123).  mage_bn_bwd_apply: dx = gamma rstd (g - s1/rows
 * - xhat s2/rows) with sums = [s1 | s2] from mode 2. */
int mage_bn_colreduce(int32_t mode, const float* x, const float* dy, const float* mask, const float* mean, const float* rstd, int64_t rows,
                      int32_t C, float* partials, int32_t n_part, void* stream);
int mage_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* residual, void* y,
                  int32_t y_dtype, int64_t rows, int32_t C, int32_t relu, void* stream);
int mage_bn_bwd_apply(const float* x, const float* dy, const float* mask, const float* mean, const float* rstd, const float* gamma,
                      const float* sums, float* dx, int64_t rows, int32_t C, void* stream);
/* Backward of mage_convt_fold_tanh: dtaps[(n, iy, ix), (ky*4+kx)*cout + co] = grad_y[n, co, oy, ox] (1 - y^2), oy = 2 iy - 1 + ky
 * (zero outside the image): the gradient of the per-input-pixel tap products of the decoder's ConvTranspose2d head.  y null: grad_y
 * is already the gradient of the pre-tanh sum. */
int mage_convt_unfold_tanh_bwd(const float* grad_y, const float* y, float* dtaps, int32_t N, int32_t IH, int32_t IW, int32_t cout, void* stream);
/* Backward of the randomness branch of MAGE.forward (mage_model.py:601-609).
 * mage_groupnorm_bwd: GroupNorm (+ residual, + ReLU / SiLU) of BasicBlock (:264-297) and of the MAGE+ head (:350-354), the row maps of
 * mage_groupnorm_act: x rows b*sample_stride_rows + row_off + r (dx is written with the same map; rows outside it are left alone),
 * dy rows b*dy_sample_stride_rows + dy_row_off + r, residual / dres packed (b*rows_per_sample + r).  stats = the forward's (mean, rstd)
 * [n_samples, groups, 2]; red = workspace of that shape; dgamma_part / dbeta_part [n_samples, C] are per-sample sums (the caller adds
 * them over samples).  Channels per group must divide 256.
 * mage_adain_bwd: ADAIN2D (:299-314) out = gamma_map * InstanceNorm(x) + beta_map: dx and dgamma_map = dout * xhat (dbeta_map = dout).
 * mage_reparam_kl_bwd: reparameterize (:569-573) + the KL term (:623): dmu = dz + c mu, dlogvar = dz eps exp(logvar/2)/2 - c (1 - exp(logvar))/2
 * with c = coef[0] = dL/dkl / B (a device scalar). */
int mage_groupnorm_bwd(const float* x, int64_t sample_stride_rows, int64_t row_off, int32_t n_samples, int32_t rows_per_sample, int32_t C,
                       int32_t groups, const float* stats, const float* gamma, const float* beta, const float* residual, int32_t act,
                       const float* dy, int64_t dy_sample_stride_rows, int64_t dy_row_off, float* red, float* dx, float* dres,
                       float* dgamma_part, float* dbeta_part, void* stream);
int mage_adain_bwd(const float* x, const float* gamma_map, const float* dout, float* dx, float* dgamma_map, int32_t B, int32_t P, int32_t C,
                   float eps, void* stream);
int mage_reparam_kl_bwd(const float* mu, const float* logvar, const float* eps, const float* dz, const float* coef, float* dmu, float* dlogvar,
                        int64_t n, void* stream);
/* Backward of mage_maxpool2 (x = the pooling input [N,H,W,C] fp32 channels-last; the gradient goes to the first maximum of each 2x2
 * window in scan order, PyTorch's tie rule) and of mage_upsample2 (dx [N,H,W,C] = the sum of each 2x2 block of dy [N,2H,2W,C]):
 * nn.MaxPool2d(2) / nn.Upsample(scale_factor=2) of the f8 VQ-VAE (vqvae_model.py:194-210) in stage-1 training. */
int mage_maxpool2_bwd(const float* x, const float* dy, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int mage_upsample2_bwd(const float* dy, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* da = d mean((a - b)^2) / da * gout[0] over the first `cols` columns (zeros in the padding columns up to ld_da): backward of mage_mse
 * (F.mse_loss of the MAGE+ latent prediction, mage_model.py:620). */
int mage_mse_bwd(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t rows, int32_t cols, const float* gout, float* da,
                 int64_t ld_da, void* stream);

/* torch.optim.Adam step (main_mage.py:121: betas (0.9, 0.98), eps 1e-6) over flat fp32 arenas; grad_scale multiplies the gradient
 * first (1 / world_size after a summing reduce-scatter). */
int mage_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int32_t step,
              float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAGE_HIP_H */
