// The MAGE_F16 instantiations of the GEMM kernels (gemm_impl.h): the bf16 kernels' schedules with f16 operands (v_mfma_f32_16x16x32_f16)
// for the single-pass f16 precision mode of the decoder stack.  A translation unit of its own so that it compiles beside gemm.hip; called by
// mage_gemm (gemm.hip) after the common argument checks.
#include "gemm_impl.h"

int mage_gemm_f16(const mage_gemm_desc* d, hipStream_t s) {
    if (const int r = try_taps8<0, true>(d, s)) return r < 0 ? r : MAGE_OK;                // row-table forms on the 8-phase kernel
    if (const int r = mage_gemm4_try(d, s)) return r < 0 ? r : MAGE_OK;                    // QKV / c_fc at full-loop sizes (gemm4.hip)
    const bool gather = d->taps_h * d->taps_w > 1 || d->stride != 1 || d->dy0 != 0 || d->dx0 != 0 || d->in_h != d->out_h || d->in_w != d->out_w;
    return gather ? launch<MAGE_F16, true>(d, s) : launch<MAGE_F16, false>(d, s);
}
