// quantize_act.hip — f32 activations -> int8 blocks, the first half of the reference CPU path's
// MUL_MAT ("quantize b rows to Q8_0 (for Q4_0/Q8_0/MXFP4) or Q8_K (K-quants)").
// Arithmetic follows the reference quantizers so that the integer dot products that follow see the
// same int8 values the CPU backend would:
//   Q8_0: gguf-py/gguf/quants.py:381-393 (bit-exact restatement of quantize_row_q8_0_ref:
//         d = amax/127, id = 1/d, q = round-half-away(x*id), d stored as f16)
//   Q8_K: [UPSTREAM-KNOWLEDGE] quantize_row_q8_K_ref (iscale = -127/max with max the FIRST element of
//         largest magnitude, q = min(127, round-half-even(iscale*x)), d = 1/iscale, 16-element bsums);
//         restated in oracle/ggml_oracle.c and tested against it bit-for-bit (tests/test_gpu_quantize.py).
// HBM-bound elementwise work: 16 B/lane loads, DPP reductions inside 8-lane groups / the wave, no LDS.
#include "dev_common.h"
#include "kernels.h"
#include "quant_core.h"

namespace mi355x {

static inline size_t pad256(size_t x) { return (x + 255) & ~(size_t) 255; }

int act_kind_for(int type_a) {
    switch (type_a) {
        case T_Q4_0: case T_Q8_0: case T_MXFP4: return T_Q8_0;
        case T_Q4_K: case T_Q5_K: case T_Q6_K:  return T_Q8_K;
        default: return -1;
    }
}

size_t act_q8_bytes(int kind, int64_t k, int64_t n) {
    const int64_t nd  = kind == T_Q8_0 ? k/32 : k/256;
    const int64_t nbs = kind == T_Q8_0 ? k/32 : k/16;
    return pad256((size_t) n*k) + pad256((size_t) n*nd*4) + pad256((size_t) n*nbs*2) + 256;
}

act_q8 act_q8_carve(void * scratch, int kind, int64_t k, int64_t n) {
    const int64_t nd  = kind == T_Q8_0 ? k/32 : k/256;
    char * p = (char *) scratch;
    act_q8 q;
    q.qs = (int8_t *) p;      p += pad256((size_t) n*k);
    q.d = (float *) p;        p += pad256((size_t) n*nd*4);
    q.bsums = (int16_t *) p;
    q.kind = kind; q.k = k; q.n = n;
    return q;
}

// 8 lanes per 32-element block, 4 consecutive floats per lane
__global__ void __launch_bounds__(256) k_quantize_q8_0(const float * __restrict__ x, int64_t n_inner, size_t stride_inner, size_t stride_outer,
                                                       int8_t * __restrict__ qs, float * __restrict__ d, int16_t * __restrict__ bsums, int64_t k) {
    const int64_t row = blockIdx.y;
    const size_t roff = (size_t)(row % n_inner)*stride_inner + (size_t)(row / n_inner)*stride_outer;
    const int64_t i0 = ((int64_t) blockIdx.x*256 + threadIdx.x)*4;
    const bool valid = i0 < k;   // k % 32 == 0, so an 8-lane group is valid or invalid as a whole
    float4v v = { 0.f, 0.f, 0.f, 0.f };
    if (valid) v = __builtin_bit_cast(float4v, ld_b128((const char *) x + roff + i0*4));   // rows may be only 4-byte aligned views
    float dd; int sum;
    const uint32_t packed = quant_frag_q8_0(v, dd, sum);
    if (valid) {
        *(uint32_t *) (qs + row*k + i0) = packed;
        if ((threadIdx.x & 7) == 0) {
            const int64_t ib = row*(k/32) + i0/32;
            d[ib] = dd;
            bsums[ib] = (int16_t) sum;
        }
    }
}

// one wave per 256-element block, 4 consecutive floats per lane
__global__ void __launch_bounds__(256) k_quantize_q8_K(const float * __restrict__ x, int64_t n_inner, size_t stride_inner, size_t stride_outer,
                                                       int8_t * __restrict__ qs, float * __restrict__ d, int16_t * __restrict__ bsums, int64_t k) {
    const int64_t row = blockIdx.y;
    const size_t roff = (size_t)(row % n_inner)*stride_inner + (size_t)(row / n_inner)*stride_outer;
    const int lane = threadIdx.x & 63;
    const int64_t blk = (int64_t) blockIdx.x*4 + (threadIdx.x >> 6);
    if (blk*256 >= k) return;   // wave-uniform
    const float4v v = __builtin_bit_cast(float4v, ld_b128((const char *) x + roff + (blk*256 + lane*4)*4));
    quant_store_chunk256<T_Q8_K>(v, (int) blk, lane, qs + row*k, d + row*(k/256), bsums + row*(k/16));
}

void quantize_act(const float * x, int64_t n_inner, size_t stride_inner, size_t stride_outer, const act_q8 & q, hipStream_t stream) {
    if (q.n == 0 || q.k == 0) return;
    const dim3 grid((unsigned)((q.k + 1023)/1024), (unsigned) q.n);
    if (q.kind == T_Q8_0) {
        hipLaunchKernelGGL(k_quantize_q8_0, grid, dim3(256), 0, stream, x, n_inner, stride_inner, stride_outer, q.qs, q.d, q.bsums, q.k);
    } else {
        hipLaunchKernelGGL(k_quantize_q8_K, grid, dim3(256), 0, stream, x, n_inner, stride_inner, stride_outer, q.qs, q.d, q.bsums, q.k);
    }
}

} // namespace mi355x
