// mage_gemm / mage_gemm_is_small: argument checks and dispatch for the fp32, bf16 and split-precision GEMMs (the kernel templates and their
// launchers live in gemm_impl.h; the f16 instantiations are compiled in gemm_f16.hip, the one-wave-per-SIMD kernel in gemm4.hip).
#include "gemm_impl.h"

int mage_gemm_f16(const mage_gemm_desc* d, hipStream_t s);     // gemm_f16.hip

#ifdef MAGE_PROBE
extern "C" int mage_debug_read_waves(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mage_probe_wave), bytes < sizeof(mage_probe_wave) ? bytes : sizeof(mage_probe_wave)) ==
                   hipSuccess
               ? 0
               : -1;
}
extern "C" int mage_debug_read_seg(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mage_probe_seg), bytes < sizeof(mage_probe_seg) ? bytes : sizeof(mage_probe_seg)) ==
                   hipSuccess
               ? 0
               : -1;
}
extern "C" int mage_debug_read(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mage_probe_buf), bytes < sizeof(mage_probe_buf) ? bytes : sizeof(mage_probe_buf)) ==
                   hipSuccess
               ? 0
               : -1;
}
#endif


extern "C" int mage_gemm_is_small(int32_t M, int32_t N, int32_t K) {
    const int dev = mage_device_index();
    MAGE_CHECK_ARG(dev >= 0 && M > 0 && N > 0 && K > 0, "mage_gemm_is_small: no current device / bad sizes");
    hipDeviceProp_t p;
    int n_cu = 256;
    static int n_cu_dev[MAGE_MAX_DEVICES] = {0};
    if (!n_cu_dev[dev]) {
        if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount >= 8) n_cu = p.multiProcessorCount & ~7;
        n_cu_dev[dev] = n_cu;
    }
    return small_shape(M, N, K, n_cu_dev[dev]) ? 1 : 0;
}

extern "C" int mage_gemm(const mage_gemm_desc* d_in, void* stream) {
    MAGE_CHECK_ARG(d_in != nullptr, "mage_gemm: null descriptor");
    mage_gemm_desc dn = *d_in;                     // defaults of the optional fields
    const bool spl = dn.dtype == MAGE_BF16X3 || dn.dtype == MAGE_F16X3;
    if (dn.ldw == 0) dn.ldw = spl ? 2 * dn.K : dn.K;
    if (dn.n_split <= 0) dn.n_split = 1;
    const mage_gemm_desc* d = &dn;
    MAGE_CHECK_ARG(d->ldw >= (spl ? 2 * d->K : d->K), "mage_gemm: ldw=%d < K=%d", d->ldw, d->K);
    MAGE_CHECK_ARG(mage_zero_page() != nullptr, "mage_gemm: mage_init() has not been called");
    const bool h16 = d->dtype == MAGE_F16;
    MAGE_CHECK_ARG(d->dtype == MAGE_F32 || d->dtype == MAGE_BF16 || h16 || spl, "mage_gemm: bad dtype %d", d->dtype);
    MAGE_CHECK_ARG(d->y_dtype == MAGE_F32 || (d->y_dtype == MAGE_BF16 && !h16) || ((spl || h16) && d->y_dtype == d->dtype),
                   "mage_gemm: bad y_dtype %d for dtype %d (fp32, or the operands' 16-bit / split type; fp32 operands may write bf16)", d->y_dtype, d->dtype);
    MAGE_CHECK_ARG(!d->residual || d->res_dtype == MAGE_F32 || (d->res_dtype == MAGE_BF16 && !h16) || (h16 && d->res_dtype == MAGE_F16),
                   "mage_gemm: bad res_dtype %d for dtype %d", d->res_dtype, d->dtype);
    MAGE_CHECK_ARG(!spl || (d->K % 64 == 0 && d->cin % 64 == 0 && d->lda >= 2 * d->cin && d->y_dtype != MAGE_BF16),
                   "mage_gemm: split-precision operands need K and cin multiples of 64, lda >= 2 cin (16-bit elements), fp32 or split output");
    MAGE_CHECK_ARG(!spl || d->y_dtype == MAGE_F32 || (d->N % 64 == 0 && d->ldy % 8 == 0 && d->ldy >= 2 * d->N && (((uintptr_t)d->Y) & 255) == 0),
                   "mage_gemm: split output needs N %% 64 == 0, ldy >= 2N (16-bit elements), Y 256-byte aligned");
    MAGE_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "mage_gemm: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
    MAGE_CHECK_ARG(d->A && d->W && d->Y, "mage_gemm: null operand");
    const int ch = d->dtype == MAGE_F32 ? 4 : 8;
    MAGE_CHECK_ARG(d->N % 8 == 0, "mage_gemm: N=%d must be a multiple of 8", d->N);
    MAGE_CHECK_ARG(d->K % ch == 0 && d->lda % ch == 0 && d->cin % ch == 0,
                   "mage_gemm: K=%d, lda=%d, cin=%d must be multiples of %d", d->K, d->lda, d->cin, ch);
    MAGE_CHECK_ARG(d->ldy % 4 == 0 && (!d->residual || d->ldr % 4 == 0), "mage_gemm: ldy/ldr must be multiples of 4");
    MAGE_CHECK_ARG((d->y_dtype != MAGE_BF16 && d->y_dtype != MAGE_F16) || d->ldy % 8 == 0, "mage_gemm: 16-bit output needs ldy a multiple of 8 (16-byte accesses)");
    MAGE_CHECK_ARG(!d->residual || d->res_dtype == MAGE_F32 || d->ldr % 8 == 0, "mage_gemm: a 16-bit residual needs ldr a multiple of 8 (16-byte accesses)");
    MAGE_CHECK_ARG(d->taps_h >= 1 && d->taps_w >= 1 && d->K == d->taps_h * d->taps_w * d->cin,
                   "mage_gemm: K=%d != taps_h*taps_w*cin = %d*%d*%d", d->K, d->taps_h, d->taps_w, d->cin);
    MAGE_CHECK_ARG(d->out_h >= 1 && d->out_w >= 1 && d->in_h >= 1 && d->in_w >= 1, "mage_gemm: bad geometry");
    MAGE_CHECK_ARG(!d->scale == !d->shift, "mage_gemm: scale and shift must be given together");
    MAGE_CHECK_ARG(!d->rowadd || (d->rowadd_div >= 1 && d->rowadd_mod >= 1), "mage_gemm: bad rowadd div/mod");
    MAGE_CHECK_ARG((((uintptr_t)d->A | (uintptr_t)d->W | (uintptr_t)d->Y) & 15) == 0, "mage_gemm: operands must be 16-byte aligned");
    const bool gather_ = d->taps_h * d->taps_w > 1 || d->stride != 1 || d->dy0 != 0 || d->dx0 != 0 || d->in_h != d->out_h || d->in_w != d->out_w;
    MAGE_CHECK_ARG(d->n_split == 1 || (!gather_ && !d->residual && !d->rowadd && !d->scale && !d->post_relu && !d->bias &&
                                       d->out_h == 1 && d->out_w >= d->M && d->act == MAGE_ACT_NONE),
                   "mage_gemm: n_split > 1 is the plain split-K form (no gather, no epilogue extras, rows not regrouped)");
    MAGE_CHECK_ARG(d->n_split == 1 || ((d->a_split_stride | d->w_split_stride) % ch == 0 && d->y_split_stride % 4 == 0 && d->ldw % ch == 0),
                   "mage_gemm: split strides / ldw must keep 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    MAGE_CHECK_ARG(!d->a_half || (d->in_h % 2 == 0 && d->in_w % 2 == 0 && d->n_split == 1), "mage_gemm: a_half needs an even in_h x in_w grid");
    MAGE_CHECK_ARG(!d->res_half || (d->residual && d->out_h > 1 && d->out_h % 2 == 0 && d->out_w % 2 == 0 && d->n_split == 1),
                   "mage_gemm: res_half needs a residual and an even out_h x out_w output plane");
    // producer: the partial sums of OUTPUT row r go to ln_part[slice][r]; consumer (ln_colsum set): it reads those of its INPUT rows 0..M-1
    MAGE_CHECK_ARG(!d->ln_part || d->ln_part_rows > (d->ln_colsum ? (int64_t)d->M - 1 : (int64_t)(d->M - 1) * d->y_mul_x + d->y_off),
                   "mage_gemm: ln_part needs ln_part_rows (rows of the slice-major partial-sum buffer) > the largest row it is indexed with");
    MAGE_CHECK_ARG(!d->a_relu || (d->dtype == MAGE_BF16 && !gather_ && !d->a_half && d->N <= 128 && d->n_split == 1 && !d->residual && !d->rowadd && !d->scale &&
                                  !d->post_relu && !d->y2 && !d->ln_part && !d->ln_stats && !d->ln_colsum && !d->head_w),
                   "mage_gemm: a_relu is a form of the bf16 plain GEMM with N <= 128 (epilogue act(acc + bias) only)");
    if (spl) {
        const bool tapsform = gather_ || d->rowadd;
        MAGE_CHECK_ARG(!d->head_w, "mage_gemm: head_w is a fusion of the bf16 padded-taps form");
        if (tapsform) {
            const int r = d->dtype == MAGE_BF16X3 ? try_taps8<1>(d, s) : try_taps8<2>(d, s);
            if (r) return r < 0 ? r : MAGE_OK;
            mage_set_error("mage_gemm: split-precision form: this convolution / row-table geometry is not eligible for the padded-taps kernel");
            return MAGE_EUNSUPPORTED;
        }
        return d->dtype == MAGE_BF16X3 ? launch_spl<1>(d, s) : launch_spl<2>(d, s);
    }
    if (h16) {                                     // the f16 instantiations (gemm_f16.hip): the generation path's forms
        MAGE_CHECK_ARG(!d->head_w && d->n_split == 1 && !d->scale && !d->post_relu && !d->res_half && !d->a_half,
                       "mage_gemm: MAGE_F16 operands do not take head_w / n_split / scale+shift / post_relu / res_half / a_half");
        return mage_gemm_f16(d, s);
    }
    if (const int r = try_taps8(d, s)) return r < 0 ? r : MAGE_OK;
    MAGE_CHECK_ARG(!d->head_w, "mage_gemm: head_w is a fusion of the bf16 padded-taps form; this geometry does not run there");
    if (const int r = mage_conv3x3_c64_try(d, s)) return r < 0 ? r : MAGE_OK;    // 64 -> 64 channel 3x3 convolutions over whole 16 x 16 tiles (conv_tile.hip)
    if (const int r = mage_gemm4_try(d, s)) return r < 0 ? r : MAGE_OK;          // the one-wave-per-SIMD kernel (gemm4.hip): QKV / c_fc at full-loop sizes
    const bool gather = d->taps_h * d->taps_w > 1 || d->stride != 1 || d->dy0 != 0 || d->dx0 != 0 || d->in_h != d->out_h ||
                        d->in_w != d->out_w || d->a_half;
    if (d->dtype == MAGE_BF16) return gather ? launch<MAGE_BF16, true>(d, s) : launch<MAGE_BF16, false>(d, s);
    return gather ? launch<MAGE_F32, true>(d, s) : launch<MAGE_F32, false>(d, s);
}
