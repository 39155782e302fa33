// mmvq.hip — quantized mat-vec for the decode step (n <= 8 activation columns), gfx950.
//
// Replaces, for this path, the CPU backend's {quantize src1 row; vec_dot per (row, col)} loop
//  for GGML_OP_MUL_MAT nodes emitted by build_lora_mm (src/llama-graph.cpp:543-567)
// and, in its _id form, GGML_OP_MUL_MAT_ID from build_lora_mm_id (src/llama-graph.cpp:569-595).
// The integer arithmetic restates ggml's generic vec_dot_*_q8_* (oracle/ggml_oracle.c); the only
// difference from the CPU result is the order of the final f32 additions.
//
// Roofline: HBM. Algorithmic bytes per launch = m * (k/blck) * type_size (weights, read once)
// + n*k*~1.1 (int8 activations, L2-resident) + m*n*4 (dst).
//
// Mapping (wave64, no cross-wave communication):
//   * a workgroup = 4 waves; each wave owns R consecutive weight rows and walks the whole K.
//   * inside a row, LPB lanes cooperate on one block so that every lane issues ONE 16-byte (Q4_K,
//     Q5_K, Q8_0, Q4_0, MXFP4) or 3 x 8-byte (Q6_K) load of packed quants per row per step and a
//     wave-wide load instruction covers 64/LPB consecutive blocks (>= 1 KiB of contiguous HBM).
//     Loads go straight to VGPRs (cdna_hip_programming.md "GEMV / M <= 16 decode weights": LDS
//     round trip is pure overhead), nontemporal where the address is 16-byte aligned.
//   * the int8 activation vector (shared by every row) is staged ONCE per workgroup into LDS for
//     n == 1 (k + k/16 bytes), or read through L1/L2 for 2 <= n <= 8.
//   * per-lane partial sums are reduced with DPP row operations + v_readlane (dev_common.h).
#include "mmvq_core.h"
#include <algorithm>

namespace mi355x {

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
struct mmvq_args {
    const char * W; size_t w_row_stride; size_t w_expert_stride;
    int64_t m, k;
    const int8_t * a_qs; const float * a_d; const int16_t * a_bs;   // [n][...]
    float * dst; size_t dst_col_stride;                              // bytes
    // MUL_MAT_ID form (ids != nullptr): blockIdx.y = pair p = t*n_used + u
    const char * ids; size_t ids_nb0, ids_nb1; int n_used; int n_b; size_t dst_nb1, dst_nb2;
    // batched form (n_batch > 1, ids == nullptr): blockIdx.y = batch b of src1 / dst; weights of batch b / r2 (ggml's broadcast); the activation image holds
    // the batches one after the other (n columns each); dst += b*dst_nb2
    int n_batch, r2; size_t w_batch_stride;
};

template <int TYPE, int NCOLS, int R, bool LDS_ACT, bool IDS>
__global__ void __launch_bounds__(256) k_mmvq(const mmvq_args p) {
    typedef mmvq_t<TYPE> T;
    constexpr int LPB = T::LPB;
    constexpr int BPW = 64/LPB;                         // blocks per wave step
    constexpr int ND  = T::ACT == T_Q8_0 ? 32 : 256;    // elements per activation scale
    constexpr int NBS = T::ACT == T_Q8_0 ? 32 : 16;     // elements per activation bsum

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int slot = lane % LPB;
    const int64_t nb = p.k / T::QK;
    const int64_t row0 = ((int64_t) blockIdx.x*4 + wave)*R;

    const char * W = p.W;
    const int8_t * a_qs = p.a_qs; const float * a_d = p.a_d; const int16_t * a_bs = p.a_bs;
    float * dst = p.dst;
    if (IDS) {
        const int pr = blockIdx.y, t = pr / p.n_used, u = pr % p.n_used;
        const int e = *(const int32_t *) (p.ids + (size_t) t*p.ids_nb1 + (size_t) u*p.ids_nb0);
        W += (size_t) e*p.w_expert_stride;
        const int64_t arow = (int64_t) t*p.n_b + (u % p.n_b);
        a_qs += arow*p.k; a_d += arow*(p.k/ND); a_bs += arow*(p.k/NBS);
        dst = (float *) ((char *) dst + (size_t) t*p.dst_nb2 + (size_t) u*p.dst_nb1);
    } else if (p.n_batch > 1) {
        const int bt = blockIdx.y;
        W += (size_t)(bt / p.r2)*p.w_batch_stride;
        const int64_t arow = (int64_t) bt*NCOLS;
        a_qs += arow*p.k; a_d += arow*(p.k/ND); a_bs += arow*(p.k/NBS);
        dst = (float *) ((char *) dst + (size_t) bt*p.dst_nb2);
    }

    act_view av[NCOLS];
    if (LDS_ACT) {
        // stage the (single) activation column: qs | d | bsums, each padded to 16 bytes
        static_assert(!LDS_ACT || NCOLS == 1, "LDS staging is for n == 1");
        const int64_t qs_b = (p.k + 15) & ~15, d_b = ((p.k/ND)*4 + 15) & ~15, bs_b = ((p.k/NBS)*2 + 15) & ~15;
        for (int64_t i = threadIdx.x*16; i < qs_b; i += 256*16) *(int4v *) (smem + i) = *(const int4v *) ((const char *) a_qs + i);
        for (int64_t i = threadIdx.x*16; i < d_b;  i += 256*16) *(int4v *) (smem + qs_b + i) = *(const int4v *) ((const char *) a_d + i);
        for (int64_t i = threadIdx.x*16; i < bs_b; i += 256*16) *(int4v *) (smem + qs_b + d_b + i) = *(const int4v *) ((const char *) a_bs + i);
        __syncthreads();
        av[0].qs = (const int8_t *) smem; av[0].d = (const float *) (smem + qs_b); av[0].bs = (const int16_t *) (smem + qs_b + d_b);
    } else {
#pragma unroll
        for (int c = 0; c < NCOLS; c++) { av[c].qs = a_qs + c*p.k; av[c].d = a_d + c*(p.k/ND); av[c].bs = a_bs + c*(p.k/NBS); }
    }

    if (row0 >= p.m) return;   // wave-uniform; after the barrier

    float acc[NCOLS][R];
#pragma unroll
    for (int c = 0; c < NCOLS; c++)
#pragma unroll
        for (int r = 0; r < R; r++) acc[c][r] = 0.0f;

    const char * rows[R];
#pragma unroll
    for (int r = 0; r < R; r++) rows[r] = W + (size_t) min(row0 + r, p.m - 1)*p.w_row_stride;   // clamp: tail rows recompute the last row, never stored

    for (int64_t ib0 = 0; ib0 < nb; ib0 += BPW) {
        const int64_t ib = ib0 + lane/LPB;
        if (ib < nb) {
            typename T::wfrag w[R];
#pragma unroll
            for (int r = 0; r < R; r++) w[r] = T::load_w(rows[r], ib, slot);
#pragma unroll
            for (int c = 0; c < NCOLS; c++) {
                const typename T::afrag a = T::load_a(av[c], ib, slot);
#pragma unroll
                for (int r = 0; r < R; r++) acc[c][r] += T::dot(w[r], a, slot);
            }
        }
    }

#pragma unroll
    for (int c = 0; c < NCOLS; c++) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float s = wave_sum(acc[c][r]);
            if (lane == 0 && row0 + r < p.m) {
                *(float *) ((char *) dst + (size_t) c*p.dst_col_stride + (size_t)(row0 + r)*4) = s;
            }
        }
    }
}

static size_t lds_bytes_for(int act_kind, int64_t k) {
    const int nd = act_kind == T_Q8_0 ? 32 : 256, nbs = act_kind == T_Q8_0 ? 32 : 16;
    return ((k + 15) & ~15) + (((k/nd)*4 + 15) & ~15) + (((k/nbs)*2 + 15) & ~15);
}


// ------------------------------------------------------------------------------------------------
// 2 <= n <= 8 columns: persistent workgroups of 8 waves (one per CU), the n quantized activation columns staged into LDS once per
// workgroup, each wave walking row pairs with a grid stride and its weight loads two k-steps ahead in a static ring of register
// sets (the structure of decode_fused.hip's single-column kernel; the simple kernel above reads the activations through L1/L2 at
// every step and waits for every load it issues: 29-58 us at n = 8 for a 4096 x 14336 matrix, tools/op_perf.py)
// ------------------------------------------------------------------------------------------------
template <int TYPE, int NCOLS>
__global__ void __launch_bounds__(512, 1) k_mmvq_cols(const mmvq_args p) {
    typedef mmvq_t<TYPE> T;
    constexpr int LPB = T::LPB, BPW = 64/LPB, R = 2, D = 2, FW = 8;
    constexpr int ND  = T::ACT == T_Q8_0 ? 32 : 256, NBS = T::ACT == T_Q8_0 ? 32 : 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = lane % LPB, ibl = lane / LPB;
    const int nb = (int)(p.k / T::QK);
    const int iters = (nb + BPW - 1)/BPW;
    const int P = (int)((p.m + R - 1)/R);
    const int stride = gridDim.x*FW;
    const int p_first = blockIdx.x*FW + wave;

    // weight prefetch first (HBM), then the activation images (L2) — the images are small and the barrier below waits for them anyway
    int p_pf = p_first, it_pf = 0;
    typename T::wfrag w[D][R];
#define MV_FETCH(d_) { \
        const bool live = p_pf < P; \
        const int pp = live ? p_pf : min(p_first, P - 1); \
        const int ibf = live ? min(it_pf*BPW + ibl, nb - 1) : 0; \
        _Pragma("unroll") for (int r = 0; r < R; r++) \
            w[d_][r] = T::load_w(p.W + (size_t) min((int64_t) pp*R + r, p.m - 1)*p.w_row_stride, ibf, slot); \
        if (++it_pf == iters) { it_pf = 0; p_pf += stride; } }
    const size_t qs_b = ((size_t) p.k + 15) & ~(size_t) 15, d_b = (((size_t) p.k/ND)*4 + 15) & ~(size_t) 15, bs_b = (((size_t) p.k/NBS)*2 + 15) & ~(size_t) 15;
    const size_t img = qs_b + d_b + bs_b;
    act_view av[NCOLS];
#pragma unroll
    for (int c = 0; c < NCOLS; c++) {
        char * base = smem + (size_t) c*img;
        const char * g_qs = (const char *) (p.a_qs + (size_t) c*p.k); const char * g_d = (const char *) (p.a_d + (size_t) c*(p.k/ND));
        const char * g_bs = (const char *) (p.a_bs + (size_t) c*(p.k/NBS));
        for (size_t i = (size_t) threadIdx.x*16; i < qs_b; i += 512*16) *(int4v *) (base + i) = ld_b128(g_qs + i);
        for (size_t i = (size_t) threadIdx.x*4;  i < d_b;  i += 512*4)  *(uint32_t *) (base + qs_b + i) = ld_u32(g_d + i);      // up to 12 bytes past the column: the next column or the region's 256-byte pad
        for (size_t i = (size_t) threadIdx.x*4;  i < bs_b; i += 512*4)  *(uint32_t *) (base + qs_b + d_b + i) = ld_u32(g_bs + i);
        av[c].qs = (const int8_t *) base; av[c].d = (const float *) (base + qs_b); av[c].bs = (const int16_t *) (base + qs_b + d_b);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int d = 0; d < D; d++) MV_FETCH(d)
    __syncthreads();

    const int my_pairs = p_first < P ? (P - 1 - p_first)/stride + 1 : 0;
    const int total = my_pairs*iters;
    int it = 0, p_cur = p_first;
    float acc[NCOLS][R];
#pragma unroll
    for (int c = 0; c < NCOLS; c++)
#pragma unroll
        for (int r = 0; r < R; r++) acc[c][r] = 0.0f;
    for (int s = 0; s < total; s += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (s + d < total) {        // wave-uniform
                const int ib = it*BPW + ibl;
                if (ib < nb) {
#pragma unroll
                    for (int c = 0; c < NCOLS; c++) {
                        const typename T::afrag a = T::load_a(av[c], ib, slot);
#pragma unroll
                        for (int r = 0; r < R; r++) acc[c][r] += T::dot(w[d][r], a, slot);
                    }
                }
                MV_FETCH(d)
                if (++it == iters) {
                    const int64_t row0 = (int64_t) p_cur*R;
#pragma unroll
                    for (int c = 0; c < NCOLS; c++) {
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            const float sum = wave_sum(acc[c][r]);
                            if (lane == 0 && row0 + r < p.m) *(float *) ((char *) p.dst + (size_t) c*p.dst_col_stride + (size_t)(row0 + r)*4) = sum;
                            acc[c][r] = 0.0f;
                        }
                    }
                    it = 0; p_cur += stride;
                }
            }
        }
    }
#undef MV_FETCH
}

template <int TYPE, int NCOLS>
static bool launch_mmvq_cols(const mmvq_args & a, int act_kind, hipStream_t stream) {
    const size_t lds = (size_t) NCOLS*lds_bytes_for(act_kind, a.k);
    if (lds > 150*1024 || a.k % 16 != 0 || a.m >= (1ll << 30)) return false;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        n_cu = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    if (!MI_LDS_LIMIT(152*1024, k_mmvq_cols<TYPE, NCOLS>)) return false;
    const int64_t pairs = (a.m + 1)/2;
    const int blocks = (int) std::min<int64_t>(n_cu, (pairs + 7)/8);
    hipLaunchKernelGGL((k_mmvq_cols<TYPE, NCOLS>), dim3((unsigned) blocks), dim3(512), lds, stream, a);
    return true;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool mul_mat_vec_q_supported(int type_a) {
    switch (type_a) {
        case T_Q4_0: case T_Q8_0: case T_Q4_K: case T_Q5_K: case T_Q6_K: case T_MXFP4: return true;
        default: return false;
    }
}

template <int TYPE, int NCOLS, bool IDS>
static void launch_mmvq_n(const mmvq_args & a, int act_kind, int64_t n_pairs, hipStream_t stream) {
    constexpr int R = 2;
    const dim3 grid((unsigned)((a.m + 4*R - 1)/(4*R)), (unsigned) n_pairs);
    if (NCOLS == 1) {
        const size_t lds = lds_bytes_for(act_kind, a.k);
        if (lds <= 64*1024) {
            hipLaunchKernelGGL((k_mmvq<TYPE, 1, R, true, IDS>), grid, dim3(256), lds, stream, a);
            return;
        }
    }
    // measured (tools/op_perf.py, m = 4096, k = 14336): faster for the 32-element block formats (Q8_0 n = 8: 57.7 -> 30.4 us, Q4_0 35.9 ->
    // 27.4) but not for the K-quants, whose n dot chains per block are VALU-bound either way (Q4_K n = 8: 29.5 -> 35.5 us with 2 waves
    // per SIMD; 16 waves per workgroup spill) — those keep the simple kernel until the int8 MFMA kernel exists
    if (NCOLS > 1 && !IDS && a.n_batch <= 1 && (TYPE == T_Q8_0 || TYPE == T_Q4_0)) {
        static int use_cols = -1;
        if (use_cols < 0) { const char * e = getenv("GGML_MI355X_MMVQ_COLS"); use_cols = e ? atoi(e) : 1; }
        if (use_cols && launch_mmvq_cols<TYPE, (NCOLS > 1 ? NCOLS : 2)>(a, act_kind, stream)) return;
    }
    hipLaunchKernelGGL((k_mmvq<TYPE, NCOLS, R, false, IDS>), grid, dim3(256), 0, stream, a);
}

template <int TYPE, bool IDS>
static void launch_mmvq_t(const mmvq_args & a, int act_kind, int64_t n, int64_t n_pairs, hipStream_t stream) {
    switch (n) {
        case 1: launch_mmvq_n<TYPE, 1, IDS>(a, act_kind, n_pairs, stream); break;
        case 2: launch_mmvq_n<TYPE, 2, IDS>(a, act_kind, n_pairs, stream); break;
        case 3: launch_mmvq_n<TYPE, 3, IDS>(a, act_kind, n_pairs, stream); break;
        case 4: launch_mmvq_n<TYPE, 4, IDS>(a, act_kind, n_pairs, stream); break;
        case 5: launch_mmvq_n<TYPE, 5, IDS>(a, act_kind, n_pairs, stream); break;
        case 6: launch_mmvq_n<TYPE, 6, IDS>(a, act_kind, n_pairs, stream); break;
        case 7: launch_mmvq_n<TYPE, 7, IDS>(a, act_kind, n_pairs, stream); break;
        case 8: launch_mmvq_n<TYPE, 8, IDS>(a, act_kind, n_pairs, stream); break;
        default: fprintf(stderr, "mmvq: n=%lld out of range\n", (long long) n); abort();
    }
}

template <bool IDS>
static void launch_mmvq(int type_a, const mmvq_args & a, int act_kind, int64_t n, int64_t n_pairs, hipStream_t stream) {
    switch (type_a) {
        case T_Q4_0:  launch_mmvq_t<T_Q4_0,  IDS>(a, act_kind, n, n_pairs, stream); break;
        case T_Q8_0:  launch_mmvq_t<T_Q8_0,  IDS>(a, act_kind, n, n_pairs, stream); break;
        case T_Q4_K:  launch_mmvq_t<T_Q4_K,  IDS>(a, act_kind, n, n_pairs, stream); break;
        case T_Q5_K:  launch_mmvq_t<T_Q5_K,  IDS>(a, act_kind, n, n_pairs, stream); break;
        case T_Q6_K:  launch_mmvq_t<T_Q6_K,  IDS>(a, act_kind, n, n_pairs, stream); break;
        case T_MXFP4: launch_mmvq_t<T_MXFP4, IDS>(a, act_kind, n, n_pairs, stream); break;
        default: fprintf(stderr, "mmvq: unsupported type %d\n", type_a); abort();
    }
}

void mul_mat_vec_q(int type_a, const void * W, size_t w_row_stride, int64_t m, int64_t k,
                   const act_q8 & act, int64_t n, float * dst, size_t dst_col_stride_bytes, hipStream_t stream) {
    if (m == 0 || n == 0) return;
    // 2..8 columns of a K-quant: the streamed kernel (mmvq_stream_cols.hip), else the int8 matrix-core kernel (mmvq_cols_mfma.hip) when the column images fit in LDS
    if (n >= 2 && mul_mat_vec_q_stream_cols(type_a, W, w_row_stride, m, k, act, n, dst, dst_col_stride_bytes, stream)) return;
    if (n >= 2 && mul_mat_vec_q_cols_mfma(type_a, W, w_row_stride, m, k, act, n, dst, dst_col_stride_bytes, stream)) return;
    mmvq_args a = {};
    a.W = (const char *) W; a.w_row_stride = w_row_stride; a.m = m; a.k = k;
    a.a_qs = act.qs; a.a_d = act.d; a.a_bs = act.bsums;
    a.dst = dst; a.dst_col_stride = dst_col_stride_bytes;
    launch_mmvq<false>(type_a, a, act.kind, n, 1, stream);
}

// the same for n_batch batches in one launch: batch b reads the weights at W + (b / r2)*w_batch_stride, columns [b*n, (b+1)*n) of the image, writes dst + b*dst_batch_stride
void mul_mat_vec_q_batched(int type_a, const void * W, size_t w_row_stride, size_t w_batch_stride, int r2, int64_t m, int64_t k,
                           const act_q8 & act, int64_t n, int64_t n_batch, float * dst, size_t dst_col_stride_bytes, size_t dst_batch_stride_bytes, hipStream_t stream) {
    if (m == 0 || n == 0 || n_batch == 0) return;
    mmvq_args a = {};
    a.W = (const char *) W; a.w_row_stride = w_row_stride; a.m = m; a.k = k;
    a.a_qs = act.qs; a.a_d = act.d; a.a_bs = act.bsums;
    a.dst = dst; a.dst_col_stride = dst_col_stride_bytes;
    a.n_batch = (int) n_batch; a.r2 = r2; a.w_batch_stride = w_batch_stride; a.dst_nb2 = dst_batch_stride_bytes;
    launch_mmvq<false>(type_a, a, act.kind, n, n_batch, stream);
}

void mul_mat_vec_q_id(int type_a, const void * W, size_t w_row_stride, size_t w_expert_stride, int64_t m, int64_t k,
                      const act_q8 & act, const int32_t * ids, size_t ids_nb0, size_t ids_nb1,
                      int64_t n_used, int64_t n_tokens, int64_t n_b,
                      float * dst, size_t dst_nb1, size_t dst_nb2, hipStream_t stream) {
    if (m == 0 || n_used == 0 || n_tokens == 0) return;
    mmvq_args a = {};
    a.W = (const char *) W; a.w_row_stride = w_row_stride; a.w_expert_stride = w_expert_stride; a.m = m; a.k = k;
    a.a_qs = act.qs; a.a_d = act.d; a.a_bs = act.bsums;
    a.dst = dst; a.dst_col_stride = 0;
    a.ids = (const char *) ids; a.ids_nb0 = ids_nb0; a.ids_nb1 = ids_nb1; a.n_used = (int) n_used; a.n_b = (int) n_b;
    a.dst_nb1 = dst_nb1; a.dst_nb2 = dst_nb2;
    launch_mmvq<true>(type_a, a, act.kind, 1, n_used*n_tokens, stream);
}

} // namespace mi355x
