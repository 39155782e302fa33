// mmq.hip — quantized mat-mul for n > MMVQ_MAX_N activation columns (prefill, pp512).
// See the kernel comment for the tiling. Until a type has an MFMA kernel the launcher falls back to
// tiling the decode mat-vec kernel over groups of 8 columns (correct; re-reads W once per group).
#include "blocks.h"
#include "dev_common.h"
#include "kernels.h"

namespace mi355x {

static act_q8 act_cols(const act_q8 & a, int64_t c0, int64_t nc) {
    const int64_t nd  = a.kind == T_Q8_0 ? a.k/32 : a.k/256;
    const int64_t nbs = a.kind == T_Q8_0 ? a.k/32 : a.k/16;
    act_q8 r = a;
    r.qs += c0*a.k; r.d += c0*nd; r.bsums += c0*nbs; r.n = nc;
    return r;
}

void mul_mat_q(int type_a, const void * W, size_t w_row_stride, int64_t m, int64_t k,
               const act_q8 & act, int64_t n, float * dst, size_t dst_col_stride_bytes, hipStream_t stream) {
    for (int64_t c0 = 0; c0 < n; c0 += MMVQ_MAX_N) {
        const int64_t nc = n - c0 < MMVQ_MAX_N ? n - c0 : MMVQ_MAX_N;
        mul_mat_vec_q(type_a, W, w_row_stride, m, k, act_cols(act, c0, nc), nc,
                      (float *) ((char *) dst + c0*dst_col_stride_bytes), dst_col_stride_bytes, stream);
    }
}

} // namespace mi355x
