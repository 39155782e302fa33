// decode_fused.hip — fused kernels for the single-token decode step (llama-bench tg).
//
// A decode graph is ~25 tiny dependent kernels per layer, each bounded by launch/boundary latency rather than by
// work (profiles/r01_a_*: 773 launches per token, ~4.7 us each). graph_compute therefore recognises the node groups
// that llm_build_llama emits next to each other (src/llama-model.cpp:5990-6104, src/llama-graph.cpp:1438-1488,
// :1283-1341, :632-774) and runs each group as ONE kernel with the same arithmetic:
//
//   RMS_NORM -> MUL(weight)                      + int8 activation quantization for the mat-vecs that follow
//   MUL_MAT(wq|wk) -> RESHAPE -> ROPE            mat-vec with the rotation in the epilogue (the wave owns rows 2i, 2i+1)
//   MUL_MAT(wq) + MUL_MAT(wk) + MUL_MAT(wv)      one grouped launch (same activation, three weight tensors)
//   SET_ROWS(k) + SET_ROWS(v)                    one KV-store launch
//   MUL_MAT(k,q) -> SOFT_MAX -> MUL_MAT(v,kq) -> PERMUTE -> CONT     one attention kernel per (head, token)
//   MUL_MAT(wo|down) -> ADD(residual)            mat-vec with the residual in the epilogue
//   MUL_MAT(up) + MUL_MAT(gate) -> GLU(swiglu)   dual mat-vec with silu(g)*u in the epilogue
//   GLU / attention output -> quantization       folded into the producer or the consumer's prologue
//
// Every fused kernel is checked against the unfused node-by-node execution (tests/test_gpu_llama_graph.py) and,
// through it, against the oracle.
#include "mmvq_core.h"
#include "quant_core.h"
#include "kv_types.h"
#include "rope_dev.h"

#include <math.h>

namespace mi355x {

static __device__ __forceinline__ float block_sum4(float v, float * sh) {   // 256 threads
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---------------------------------------------------------------------------------------------------------------
// RMS_NORM * w  -> f32 row + quantized row           (src/llama-graph.cpp:597-630 + the MUL_MAT's activation quantizer)
// one workgroup (256 threads) per row; ne0 % 256 == 0 (Q8_K) or % 32 == 0 (Q8_0, chunks of 256 still: ne0 % 256 == 0 required)
// ---------------------------------------------------------------------------------------------------------------
// CH = 256-element chunks per wave held in registers (ne0 <= CH*1024): x and w are read ONCE, all loads issued up front.
template <int ACT, int CH>
__global__ void __launch_bounds__(256) k_rms_norm_mul_quant(const float * __restrict__ x, size_t x_stride, const float * __restrict__ w,
                                                            float * __restrict__ y, size_t y_stride, int8_t * qs, float * d, int16_t * bs,
                                                            int ne0, float eps) {
    __shared__ float sh[4];
    const int row = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float * xr = (const float *) ((const char *) x + (size_t) row*x_stride);
    float * yr = (float *) ((char *) y + (size_t) row*y_stride);
    const int nchunk = ne0/256;
    float4v xv[CH], wv[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) {
        const int c = min(wave + 4*i, nchunk - 1);
        xv[i] = *(const float4v *) (xr + c*256 + lane*4);
        wv[i] = *(const float4v *) (w + c*256 + lane*4);
    }
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < CH; i++) if (wave + 4*i < nchunk) ss += xv[i].x*xv[i].x + xv[i].y*xv[i].y + xv[i].z*xv[i].z + xv[i].w*xv[i].w;
    ss = block_sum4(ss, sh);
    const float scale = 1.0f/sqrtf(ss/(float) ne0 + eps);
    constexpr int ND = ACT == T_Q8_0 ? 32 : 256, NBS = ACT == T_Q8_0 ? 32 : 16;
    int8_t * qr = qs + (size_t) row*ne0; float * dr = d + (size_t) row*(ne0/ND); int16_t * br = bs + (size_t) row*(ne0/NBS);
#pragma unroll
    for (int i = 0; i < CH; i++) {
        const int c = wave + 4*i;
        if (c < nchunk) {   // wave-uniform
            float4v v = xv[i];
            v.x = (v.x*scale)*wv[i].x; v.y = (v.y*scale)*wv[i].y; v.z = (v.z*scale)*wv[i].z; v.w = (v.w*scale)*wv[i].w;   // RMS_NORM then MUL: two roundings, as unfused
            *(float4v *) (yr + c*256 + lane*4) = v;
            quant_store_chunk256<ACT>(v, c, lane, qr, dr, br);
        }
    }
}

// NOTE on summation order: sum(x^2) is accumulated per lane over its chunks, then DPP/LDS-reduced — a different order from
// elem.hip's k_rms_norm (which strides by thread); both are f32 sums of the same terms (relative difference ~1e-7).
template <int ACT>
static void launch_rms_norm_mul_quant(const float * x, size_t x_stride, const float * w, float * y, size_t y_stride, const act_q8 & q,
                                      int64_t ne0, int64_t nrows, float eps, hipStream_t stream) {
    const dim3 g((unsigned) nrows), b(256);
    const int64_t ch = (ne0/256 + 3)/4;
    if      (ch <= 1) hipLaunchKernelGGL((k_rms_norm_mul_quant<ACT, 1>), g, b, 0, stream, x, x_stride, w, y, y_stride, q.qs, q.d, q.bsums, (int) ne0, eps);
    else if (ch <= 2) hipLaunchKernelGGL((k_rms_norm_mul_quant<ACT, 2>), g, b, 0, stream, x, x_stride, w, y, y_stride, q.qs, q.d, q.bsums, (int) ne0, eps);
    else if (ch <= 4) hipLaunchKernelGGL((k_rms_norm_mul_quant<ACT, 4>), g, b, 0, stream, x, x_stride, w, y, y_stride, q.qs, q.d, q.bsums, (int) ne0, eps);
    else              hipLaunchKernelGGL((k_rms_norm_mul_quant<ACT, 8>), g, b, 0, stream, x, x_stride, w, y, y_stride, q.qs, q.d, q.bsums, (int) ne0, eps);
}

bool rms_norm_mul_quant_supported(int64_t ne0) { return ne0 % 256 == 0 && ne0 <= 8192; }

void rms_norm_mul_quant(const float * x, size_t x_stride, const float * w, float * y, size_t y_stride, const act_q8 & q,
                        int64_t ne0, int64_t nrows, float eps, hipStream_t stream) {
    if (q.kind == T_Q8_0) launch_rms_norm_mul_quant<T_Q8_0>(x, x_stride, w, y, y_stride, q, ne0, nrows, eps, stream);
    else                  launch_rms_norm_mul_quant<T_Q8_K>(x, x_stride, w, y, y_stride, q, ne0, nrows, eps, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// SWIGLU(split) -> f32 row + quantized row           (src/llama-graph.cpp:691 + the down-projection's quantizer)
// ---------------------------------------------------------------------------------------------------------------
template <int ACT>
__global__ void __launch_bounds__(256) k_swiglu_quant(const float * __restrict__ g, size_t g_stride, const float * __restrict__ u, size_t u_stride,
                                                      float * __restrict__ y, size_t y_stride, int8_t * qs, float * d, int16_t * bs, int ne0) {
    const int row = blockIdx.y, lane = threadIdx.x & 63;
    const int c = blockIdx.x*4 + (threadIdx.x >> 6);
    if (c*256 >= ne0) return;
    const float4v a = *(const float4v *) ((const char *) g + (size_t) row*g_stride + (size_t)(c*256 + lane*4)*4);
    const float4v b = *(const float4v *) ((const char *) u + (size_t) row*u_stride + (size_t)(c*256 + lane*4)*4);
    float4v v;
    v.x = (a.x/(1.0f + expf(-a.x)))*b.x; v.y = (a.y/(1.0f + expf(-a.y)))*b.y;
    v.z = (a.z/(1.0f + expf(-a.z)))*b.z; v.w = (a.w/(1.0f + expf(-a.w)))*b.w;
    *(float4v *) ((char *) y + (size_t) row*y_stride + (size_t)(c*256 + lane*4)*4) = v;
    constexpr int ND = ACT == T_Q8_0 ? 32 : 256, NBS = ACT == T_Q8_0 ? 32 : 16;
    quant_store_chunk256<ACT>(v, c, lane, qs + (size_t) row*ne0, d + (size_t) row*(ne0/ND), bs + (size_t) row*(ne0/NBS));
}

void swiglu_quant(const float * g, size_t g_stride, const float * u, size_t u_stride, float * y, size_t y_stride, const act_q8 & q,
                  int64_t ne0, int64_t nrows, hipStream_t stream) {
    const dim3 grid((unsigned)((ne0/256 + 3)/4), (unsigned) nrows);
    if (q.kind == T_Q8_0) hipLaunchKernelGGL((k_swiglu_quant<T_Q8_0>), grid, dim3(256), 0, stream, g, g_stride, u, u_stride, y, y_stride, q.qs, q.d, q.bsums, (int) ne0);
    else                  hipLaunchKernelGGL((k_swiglu_quant<T_Q8_K>), grid, dim3(256), 0, stream, g, g_stride, u, u_stride, y, y_stride, q.qs, q.d, q.bsums, (int) ne0);
}

// ---------------------------------------------------------------------------------------------------------------
// KV store: SET_ROWS(k_cache, k_cur, k_idxs) and SET_ROWS(v_view[1,N], v_cur[1,N], v_idxs) in one launch
// (src/llama-kv-cache-unified.cpp:1123,1157-1167). f32 -> f16.
// ---------------------------------------------------------------------------------------------------------------
struct kv_store_args {
    const float * k_src; size_t k_src_nb1; const int64_t * k_idx; uint16_t * k_dst; size_t k_dst_nb1; int k_ne0, k_rows;
    const float * v_src; const int64_t * v_idx; uint16_t * v_dst; int v_n;
};
__global__ void __launch_bounds__(256) k_kv_store(const kv_store_args p) {
    const int i = blockIdx.x*256 + threadIdx.x;
    const int nk = p.k_ne0*p.k_rows;
    if (i < nk) {
        const int r = i / p.k_ne0, c = i - r*p.k_ne0;
        const float v = *(const float *) ((const char *) p.k_src + (size_t) r*p.k_src_nb1 + (size_t) c*4);
        *(uint16_t *) ((char *) p.k_dst + (size_t) p.k_idx[r]*p.k_dst_nb1 + (size_t) c*2) = f32_to_f16_bits(v);
    } else if (i - nk < p.v_n) {
        const int e = i - nk;
        p.v_dst[p.v_idx[e]] = f32_to_f16_bits(p.v_src[e]);
    }
}
void kv_store_f16(const float * k_src, size_t k_src_nb1, const int64_t * k_idx, void * k_dst, size_t k_dst_nb1, int64_t k_ne0, int64_t k_rows,
                  const float * v_src, const int64_t * v_idx, void * v_dst, int64_t v_n, hipStream_t stream) {
    kv_store_args a = { k_src, k_src_nb1, k_idx, (uint16_t *) k_dst, k_dst_nb1, (int) k_ne0, (int) k_rows, v_src, v_idx, (uint16_t *) v_dst, (int) v_n };
    const int64_t n = k_ne0*k_rows + v_n;
    hipLaunchKernelGGL(k_kv_store, dim3((unsigned)((n + 255)/256)), dim3(256), 0, stream, a);
}

// ---------------------------------------------------------------------------------------------------------------
// attention for a few query tokens over the f16 KV cache (build_attn_mha, no-FA branch, src/llama-graph.cpp:1283-1330):
//   kq = K.q (f32) ; p = softmax(kq*scale + mask [, sink]) ; out = V^T.p ; written in cont_2d layout [hd*n_head, T]
// one workgroup per (head, token); K rows are [hd] f16 contiguous, V is the transposed cache (rows over cells).
// ---------------------------------------------------------------------------------------------------------------
struct attn_args {
    const char * q; size_t q_nb1, q_nb2;                // q [hd, T, n_head] f32 (permuted view): nb1 = token stride, nb2 = head stride
    const char * k; size_t k_nb1, k_nb2;                // k [hd, n_kv, n_head_kv] f16: nb1 = cell stride, nb2 = head stride
    const char * v; size_t v_nb1, v_nb2;                // v [n_kv, hd, n_head_kv] f16: nb1 = dim stride (row over cells), nb2 = head stride
    const char * mask; size_t m_nb1; int mask_f16;      // mask [n_kv, T_pad]
    const float * sinks;
    float * dst; size_t dst_nb1;                        // [hd*n_head, T]
    int n_kv, n_head, n_head_kv, T;
    float scale;
    // long contexts: blockIdx.z splits the cells into nsplit ranges of kv_chunk (a multiple of 32); each workgroup writes its range's
    // soft_max-weighted V sum (normalised within the range) + its max and denominator to part[(t*n_head + h)*nsplit + z][HD + 2];
    // k_attn_merge combines them (and adds the sink). nsplit = 1: the kernel finishes the row itself.
    int kv_chunk, nsplit; float * part;
    int live_scan;      // row-major V, one workgroup per head: stop at the last unmasked cell
    // FLASH_ATTN_EXT's other two parameters (ggml/src/ggml-cpu/ops.cpp ggml_compute_forward_flash_attn_ext_f16): logit_softcap != 0: the scaled score
    // (scale already divided by it on the host) goes through softcap * tanh(.); max_bias > 0: ALiBi — head h's mask values are multiplied by its slope
    float softcap, max_bias, m0, m1; int n_head_log2;
};
static __device__ __forceinline__ float attn_slope(float max_bias, float m0, float m1, int n_head_log2, int h) {
    return max_bias > 0.0f ? (h < n_head_log2 ? powf(m0, (float)(h + 1)) : powf(m1, (float)(2*(h - n_head_log2) + 1))) : 1.0f;
}

static __device__ __forceinline__ float dot8_f16_f32(const int4v kv, const float4v a, const float4v b) {
    const uint32_t k0 = (uint32_t) kv.x, k1 = (uint32_t) kv.y, k2 = (uint32_t) kv.z, k3 = (uint32_t) kv.w;
    float acc;
    acc  = f16_bits_to_f32((uint16_t) k0)*a.x + f16_bits_to_f32((uint16_t)(k0 >> 16))*a.y;
    acc += f16_bits_to_f32((uint16_t) k1)*a.z + f16_bits_to_f32((uint16_t)(k1 >> 16))*a.w;
    acc += f16_bits_to_f32((uint16_t) k2)*b.x + f16_bits_to_f32((uint16_t)(k2 >> 16))*b.y;
    acc += f16_bits_to_f32((uint16_t) k3)*b.z + f16_bits_to_f32((uint16_t)(k3 >> 16))*b.w;
    return acc;
}

// Latency is the enemy here (a few hundred KB, one dependent chain per workgroup): every phase issues its independent
// 16-byte loads in batches of U before touching the data.
// VT: V is the transposed cache [n_kv, hd] (rows over cells; the graph without flash attention). !VT: V rows are cells [hd, n_kv]
// (FLASH_ATTN_EXT, src/llama-graph.cpp:1245-1265): v_nb1 is then the cell stride.
// KQ: the K cache is Q8_0 (-ctk q8_0): a head's row is HD/32 blocks of {f16 d, 32 int8}; a lane's 8 elements are 8 bytes of one block
// KT / VY: the element types of the K and V rows (kv_types.h). The transposed-V form (VT) takes F16 or Q8_0 K and F16 V only.
template <int HD, bool VT = true, int KT = T_F16, int VY = T_F16>
__global__ void __launch_bounds__(256) k_attn_decode(const attn_args p) {
    constexpr bool KQ = KT == T_Q8_0;
    static_assert(!VT || ((KT == T_F16 || KT == T_Q8_0) && VY == T_F16), "transposed V: f16 V, f16 or Q8_0 K");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float * s = (float *) smem;                          // [n_kv] scores -> probabilities
    __shared__ float sh[4];
    const int h = blockIdx.x, t = blockIdx.y;
    const int hk = h/(p.n_head/p.n_head_kv);
    const bool split = p.nsplit > 1;
    const int kv_lo = split ? (int) blockIdx.z*p.kv_chunk : 0;
    int kv_n = split ? min(p.n_kv - kv_lo, p.kv_chunk) : p.n_kv;   // this workgroup's cells
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPC = HD/8;                            // lanes per K row (8 f16 = 16 B each)
    constexpr int CPW = 64/LPC;                          // K rows per wave step
    constexpr int U = 4;
    const int sub = lane % LPC, cw = lane / LPC;
    // q: written by the launch before on other XCDs, i.e. the longest wait of this launch — asked for before anything else (the live scan below waits for the mask)
    const float * qp = (const float *) (p.q + (size_t) t*p.q_nb1 + (size_t) h*p.q_nb2) + sub*8;
    const float4v q0_ld = *(const float4v *) qp, q1_ld = *(const float4v *) (qp + 4);
    if (!VT && !split && p.live_scan && p.mask) {
        // flash attention pads the cache view to 256 cells; everything behind the last unmasked cell is dead weight for K, the soft_max and V: find that cell
        // first (one mask value per thread and trip) and walk only up to it (a multiple of 8 cells, at least 8)
        __shared__ int live_w[4];
        const char * mr = p.mask + (size_t) blockIdx.y*p.m_nb1;
        int last = -1;
        for (int j = threadIdx.x; j < kv_n; j += 256) {
            const float mv = p.mask_f16 ? f16_bits_to_f32(*(const uint16_t *) (mr + (size_t) j*2)) : *(const float *) (mr + (size_t) j*4);
            if (mv != -INFINITY) last = j;
        }
        const float lw = wave_max((float) last);
        if ((threadIdx.x & 63) == 0) live_w[threadIdx.x >> 6] = (int) lw;
        __syncthreads();
        const int lm = max(max(live_w[0], live_w[1]), max(live_w[2], live_w[3]));
        kv_n = min(kv_n, max(8, (lm + 8) & ~7));
        __syncthreads();
    }

    // ---- scores: s[j] = scale * K[j].q + mask[j] ----
    if (!KQ) {
        // the no-flash-attention graph's K.q is MUL_MAT(F16 cache, F32 q): the CPU backend converts q to the F16 operand type first (vec_dot_type of
        // F16, tests/test-quantize-fns.cpp:82-99) — followed here, so that the logits match the CPU reference and not just the exact product
        // (round 3: 8192-position perplexity statistics put the f32-q kernel 1.4e-3 in ln PPL from the CPU arithmetic, all of it this rounding).
        // FLASH_ATTN_EXT on the CPU converts q the same way (its K operand type), so both forms of the kernel do
        // (applied inside the loop, BEHIND each batch of K requests: done here, the wait for q — cold, written by the launch before — came first and the K rows were
        // only asked for after q had arrived: two memory round trips in a row on a 5 us launch)
#define MI_R16(x_) x_ = f16_bits_to_f32(f32_to_f16_bits(x_))
    }
    const char * kbase = p.k + (size_t) hk*p.k_nb2 + (KT == T_F16 ? sub*16 : KQ ? (sub >> 2)*34 : 0) + (size_t) kv_lo*p.k_nb1;
    const bool alibi = p.max_bias > 0.0f;
    const float slope = attn_slope(p.max_bias, p.m0, p.m1, p.n_head_log2, h);
    const char * mrow = p.mask ? p.mask + (size_t) t*p.m_nb1 + (size_t) kv_lo*(p.mask_f16 ? 2 : 4) : nullptr;
    // transposed V: the first 128 cells' worth of every lane's V rows is requested NOW, next to q and K — the soft_max in between does
    // not need them and the loads do not need the soft_max (one memory round trip less on the chain for n_kv <= 128: tg128)
    constexpr int NGP = VT ? HD/16 : 1;
    int4v vpre[NGP];
    if (VT) {
        const int l16p = lane & 15, rwp = lane >> 4;
        const char * vb0 = p.v + (size_t) hk*p.v_nb2 + (size_t)(wave*4 + rwp)*p.v_nb1 + (size_t) kv_lo*2;
        const int c0 = min(l16p, max((kv_n >> 3) - 1, 0));
#pragma unroll
        for (int g = 0; g < NGP; g++) vpre[g] = ld_b128(vb0 + (size_t)(g*16)*p.v_nb1 + (size_t) c0*16);
    }
    float mx = (p.sinks && !split) ? p.sinks[h] : -INFINITY;
    for (int j0 = wave*CPW + cw; j0 < kv_n; j0 += 4*CPW*U) {
        int4v kreg[U]; float mreg[U]; uint32_t mraw[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = min(j0 + u*4*CPW, kv_n - 1);
            if (KQ) {     // .x, .y: the lane's 8 quants; .z: the block's scale (blocks are 34 bytes: 2-byte aligned loads)
                const int2v qq = ld_b64(kbase + (size_t) j*p.k_nb1 + 2 + (sub & 3)*8);
                kreg[u] = int4v{ qq.x, qq.y, (int) ld_u16(kbase + (size_t) j*p.k_nb1), 0 };
            } else if (KT == T_F16)
            kreg[u] = *(const int4v *) (kbase + (size_t) j*p.k_nb1);
            else kreg[u] = kv_raw8<KT>(kbase + (size_t) j*p.k_nb1, sub);
            // (the mask value is only REQUESTED here — raw bits, converted below: converting an f16 mask value on the spot put a wait for every outstanding load
            // behind each of the U row requests, one memory round trip per cell instead of one per batch: ISA of the -fa 1 form)
            mraw[u] = 0;
            if (mrow) { if (p.mask_f16) mraw[u] = *(const uint16_t *) (mrow + (size_t) j*2); else mraw[u] = *(const uint32_t *) (mrow + (size_t) j*4); }
        }
#pragma unroll
        for (int u = 0; u < U; u++) mreg[u] = !mrow ? 0.0f : p.mask_f16 ? f16_bits_to_f32((uint16_t) mraw[u]) : __builtin_bit_cast(float, mraw[u]);
        float4v q0 = q0_ld, q1 = q1_ld;
        asm volatile("" : "+v"(q0), "+v"(q1));      // (opaque: the conversion below is loop-invariant and would be hoisted back in front of the K requests)
        if (!KQ) { MI_R16(q0.x); MI_R16(q0.y); MI_R16(q0.z); MI_R16(q0.w); MI_R16(q1.x); MI_R16(q1.y); MI_R16(q1.z); MI_R16(q1.w); }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = j0 + u*4*CPW;
            float acc;
            if (KQ) {
                const int a0 = kreg[u].x, a1 = kreg[u].y;
                acc = ((float)(int8_t) a0*q0.x + (float)(int8_t)(a0 >> 8)*q0.y) + ((float)(int8_t)(a0 >> 16)*q0.z + (float)(a0 >> 24)*q0.w)
                    + ((float)(int8_t) a1*q1.x + (float)(int8_t)(a1 >> 8)*q1.y) + ((float)(int8_t)(a1 >> 16)*q1.z + (float)(a1 >> 24)*q1.w);
                acc *= f16_bits_to_f32((uint16_t) kreg[u].z);
            } else if (KT == T_F16) acc = dot8_f16_f32(kreg[u], q0, q1);
            else {
                float f[8]; kv_cvt8<KT>(kreg[u], sub, f);
                acc  = f[0]*q0.x + f[1]*q0.y; acc += f[2]*q0.z + f[3]*q0.w; acc += f[4]*q1.x + f[5]*q1.y; acc += f[6]*q1.z + f[7]*q1.w;
            }
            acc += dpp_f<0xB1>(acc); acc += dpp_f<0x4E>(acc); acc += dpp_f<0x141>(acc);   // sum over the LPC lanes of the row
            if (LPC == 16) acc += dpp_f<0x140>(acc);
            if (j < kv_n) {
                float v;
                if (p.softcap != 0.0f || alibi) { v = acc*p.scale; if (p.softcap != 0.0f) v = p.softcap*tanhf(v); v += slope*mreg[u]; }
                else v = acc*p.scale + mreg[u];
                if (sub == 0) s[j] = v;
                mx = fmaxf(mx, v);
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) sh[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    float sum = 0.0f;
    const float mxs = mx == -INFINITY ? 0.0f : mx;      // a range whose cells are all masked (split only): every e = 0
    for (int j = threadIdx.x; j < kv_n; j += 256) { const float e = expf(s[j] - mxs); s[j] = e; sum += e; }
    sum = block_sum4(sum, sh);
    if (p.sinks && !split) sum += expf(p.sinks[h] - mx);
    const float inv = sum > 0.0f ? 1.0f/sum : 0.0f;
    // the unfused SOFT_MAX normalises before V.p; and V.p is MUL_MAT(F16 cache, F32 p): the CPU converts p to F16 like q above (one range only:
    // a range of a split cache does not know the denominator yet)
    // (p = e / sum, a division like the CPU's soft_max — not e * (1 / sum) — so that the value the f16 rounding sees is the CPU's)
    if (VT && !split) { for (int j = threadIdx.x; j < kv_n; j += 256) { float pj = sum > 0.0f ? s[j]/sum : 0.0f; MI_R16(pj); s[j] = pj; } }
    else for (int j = threadIdx.x; j < kv_n; j += 256) s[j] *= inv;
    __syncthreads();
    // where the result goes: the output row, or this range's slot of the partial buffer (+ its max and denominator)
    float * orow = split ? p.part + ((size_t)(t*p.n_head + h)*p.nsplit + blockIdx.z)*(HD + 2) : (float *) ((char *) p.dst + (size_t) t*p.dst_nb1) + (size_t) h*HD;
    if (split && threadIdx.x == 0) { orow[HD] = mx; orow[HD + 1] = sum; }

    if (!VT) {
        // ---- out[d] = sum_j p[j]*V[j][d]: a thread owns 8 dims of every NGR-th cell; partial sums meet in LDS ----
        constexpr int DCH = HD/8, NGR = 256/DCH, UV = 4;
        float * red = (float *) (smem + (((size_t)(split ? p.kv_chunk : p.n_kv)*4 + 15) & ~(size_t) 15));     // [NGR][HD]
        const int dch = threadIdx.x % DCH, cg = threadIdx.x / DCH;
        const char * vb = p.v + (size_t) hk*p.v_nb2 + (VY == T_F16 ? (size_t) dch*16 : 0) + (size_t) kv_lo*p.v_nb1;
        float a8[8] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
        for (int j0 = cg; j0 < kv_n; j0 += NGR*UV) {
            int4v vr[UV]; float pj[UV];
#pragma unroll
            for (int u = 0; u < UV; u++) {
                const int j = min(j0 + u*NGR, kv_n - 1);
                if (VY == T_F16) vr[u] = *(const int4v *) (vb + (size_t) j*p.v_nb1);
                else vr[u] = kv_raw8<VY>(vb + (size_t) j*p.v_nb1, dch);
                pj[u] = j0 + u*NGR < kv_n ? s[j] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < UV; u++) {
                if (VY != T_F16) {
                    float f[8]; kv_cvt8<VY>(vr[u], dch, f);
#pragma unroll
                    for (int i = 0; i < 8; i++) a8[i] += pj[u]*f[i];
                    continue;
                }
                const uint32_t w0 = (uint32_t) vr[u].x, w1 = (uint32_t) vr[u].y, w2 = (uint32_t) vr[u].z, w3 = (uint32_t) vr[u].w;
                a8[0] += pj[u]*f16_bits_to_f32((uint16_t) w0); a8[1] += pj[u]*f16_bits_to_f32((uint16_t)(w0 >> 16));
                a8[2] += pj[u]*f16_bits_to_f32((uint16_t) w1); a8[3] += pj[u]*f16_bits_to_f32((uint16_t)(w1 >> 16));
                a8[4] += pj[u]*f16_bits_to_f32((uint16_t) w2); a8[5] += pj[u]*f16_bits_to_f32((uint16_t)(w2 >> 16));
                a8[6] += pj[u]*f16_bits_to_f32((uint16_t) w3); a8[7] += pj[u]*f16_bits_to_f32((uint16_t)(w3 >> 16));
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) red[cg*HD + dch*8 + i] = a8[i];
        __syncthreads();
        if (threadIdx.x < HD) {
            float r = 0.0f;
            for (int gq = 0; gq < NGR; gq++) r += red[gq*HD + threadIdx.x];
            orow[threadIdx.x] = r;
        }
        return;
    }
    // ---- out[d] = sum_j V[d][j]*p[j]: 16 lanes per V row, 4 rows per wave, HD/16 row groups per workgroup ----
    constexpr int NG = HD/16;                            // row groups: d = g*16 + wave*4 + rw
    const int l16 = lane & 15, rw = lane >> 4;
    const char * vbase = p.v + (size_t) hk*p.v_nb2 + (size_t)(wave*4 + rw)*p.v_nb1 + (size_t) kv_lo*2;
    float acc[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) acc[g] = 0.0f;
    const int nchunk = kv_n >> 3;                        // n_kv % 8 == 0 (and kv_chunk % 32 == 0)
    for (int c = l16; c < nchunk; c += 16) {
        int4v vreg[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) vreg[g] = c == l16 ? vpre[g] : ld_b128(vbase + (size_t)(g*16)*p.v_nb1 + (size_t) c*16);
        const float4v p0 = *(const float4v *) (s + c*8), p1 = *(const float4v *) (s + c*8 + 4);
#pragma unroll
        for (int g = 0; g < NG; g++) acc[g] += dot8_f16_f32(vreg[g], p0, p1);
    }
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const float r = row16_sum(acc[g]);
        if (l16 == 0) orow[g*16 + wave*4 + rw] = r;
    }
}

// combine the cell ranges of one (token, head): out = sum_i w_i o_i / (sum_i w_i [+ e^(sink - M)]), w_i = l_i e^(m_i - M), M = max(m_i [, sink])
template <int HD>
__global__ void __launch_bounds__(HD) k_attn_merge(const attn_args p) {
    const int h = blockIdx.x, t = blockIdx.y, d = threadIdx.x;
    const float * pr = p.part + (size_t)(t*p.n_head + h)*p.nsplit*(HD + 2);
    float M = p.sinks ? p.sinks[h] : -INFINITY;
    for (int i = 0; i < p.nsplit; i++) M = fmaxf(M, pr[(size_t) i*(HD + 2) + HD]);
    float den = p.sinks ? expf(p.sinks[h] - M) : 0.0f, acc = 0.0f;
    for (int i = 0; i < p.nsplit; i++) {
        const float mi = pr[(size_t) i*(HD + 2) + HD], li = pr[(size_t) i*(HD + 2) + HD + 1];
        const float w = li > 0.0f ? li*expf(mi - M) : 0.0f;
        den += w; acc += w*pr[(size_t) i*(HD + 2) + d];
    }
    *(float *) ((char *) p.dst + (size_t) t*p.dst_nb1 + (size_t)(h*HD + d)*4) = acc/den;
}

// without a partial buffer the whole row of scores must fit the workgroup's LDS
bool attn_decode_supported(int64_t head_dim, int64_t n_kv) { return (head_dim == 128 || head_dim == 64) && n_kv % 8 == 0 && n_kv*4 <= 48*1024; }
// measured (tg1024 / --fa 1 tg128): the transposed-V kernel gains from 384-512 cells on, the row-V (flash attention) one already at 256
static int64_t attn_split_min(bool v_trans = true) {
    static const int64_t v = getenv("GGML_MI355X_ATTN_SPLIT_MIN") ? atoll(getenv("GGML_MI355X_ATTN_SPLIT_MIN")) : 0;
    return v ? (v < 256 ? 256 : v) : (v_trans ? 384 : 256);
}
bool attn_decode_supported_split(int64_t head_dim, int64_t n_kv) { return (head_dim == 128 || head_dim == 64) && n_kv >= attn_split_min(true) && n_kv % 32 == 0; }
size_t attn_decode_part_bytes(int64_t head_dim, int64_t n_kv, int64_t n_head, int64_t T) {     // 0: no split for this shape
    if (n_kv < attn_split_min(false) || n_kv % 32 != 0) return 0;
    const int64_t ns = (n_kv + 127)/128 < 32 ? (n_kv + 127)/128 : 32;
    return (size_t) T*n_head*ns*(head_dim + 2)*4;
}

// which (K, V) type pairs the row-major-V (flash attention) kernel reads directly; the others go through kv_to_f16 first
bool attn_decode_kv_types_fused(int k_type, int v_type) {
    return (k_type == v_type && (k_type == T_F16 || k_type == T_Q8_0 || k_type == T_Q4_0 || k_type == T_BF16)) || (k_type == T_Q8_0 && (v_type == T_F16 || v_type == T_Q4_0));
}
// ALiBi slopes (ggml_compute_forward_flash_attn_ext_f16 / soft_max: m0 = 2^(-max_bias / n), m1 = 2^(-max_bias / 2 / n), n = 2^floor(log2 n_head))
void attn_alibi(float max_bias, int64_t n_head, float & m0, float & m1, int & n_head_log2) {
    n_head_log2 = 1; while (2*n_head_log2 <= (int) n_head) n_head_log2 *= 2;
    m0 = powf(2.0f, -max_bias/(float) n_head_log2); m1 = powf(2.0f, -(max_bias/2.0f)/(float) n_head_log2);
}

void attn_decode(const void * q, size_t q_nb1, size_t q_nb2, const void * k, size_t k_nb1, size_t k_nb2, const void * v, size_t v_nb1, size_t v_nb2,
                 const void * mask, size_t m_nb1, bool mask_f16, const float * sinks, float * dst, size_t dst_nb1,
                 int64_t head_dim, int64_t n_kv, int64_t n_head, int64_t n_head_kv, int64_t T, float scale, hipStream_t stream, bool v_trans,
                 float * part, size_t part_bytes, bool k_q8_0, const attn_extra * ex) {
    attn_args a = { (const char *) q, q_nb1, q_nb2, (const char *) k, k_nb1, k_nb2, (const char *) v, v_nb1, v_nb2,
                    (const char *) mask, m_nb1, mask_f16 ? 1 : 0, sinks, dst, dst_nb1, (int) n_kv, (int) n_head, (int) n_head_kv, (int) T, scale, 0, 1, nullptr };
    a.softcap = 0.0f; a.max_bias = 0.0f; a.m0 = a.m1 = 1.0f; a.n_head_log2 = 1;
    int kt = k_q8_0 ? T_Q8_0 : T_F16, vy = T_F16;
    if (ex) {
        attn_alibi(ex->max_bias, n_head, a.m0, a.m1, a.n_head_log2);
        a.softcap = ex->softcap; a.max_bias = ex->max_bias;
        if (ex->softcap != 0.0f) a.scale = scale/ex->softcap;
        if (ex->k_type) kt = ex->k_type;
        if (ex->v_type) vy = ex->v_type;
    }
    if (v_trans ? !((kt == T_F16 || kt == T_Q8_0) && vy == T_F16) : !attn_decode_kv_types_fused(kt, vy)) { fprintf(stderr, "attn_decode: KV types (%d, %d) have no kernel\n", kt, vy); abort(); }
    // long contexts: one workgroup per (head, token) walks every cell alone — 32 workgroups on 256 CUs. With a partial buffer the cells
    // are split into ranges of >= 256 (at most 32 ranges) that run side by side and a small second kernel merges them.
    static const bool split_on = !getenv("GGML_MI355X_ATTN_SPLIT") || atoi(getenv("GGML_MI355X_ATTN_SPLIT")) != 0;
    const size_t need = attn_decode_part_bytes(head_dim, n_kv, n_head, T);
    // row-major V (flash attention) at exactly the padding size: most cells of a short context are masked padding — one workgroup per head that stops at the last
    // live cell beats two ranges + a merge launch there (GGML_MI355X_ATTN_LIVE_SCAN=0: the split as before)
    static const bool live_on = !getenv("GGML_MI355X_ATTN_LIVE_SCAN") || atoi(getenv("GGML_MI355X_ATTN_LIVE_SCAN")) != 0;
    const bool live_scan = live_on && !v_trans && mask && n_kv == 256 && T == 1;
    a.live_scan = live_scan ? 1 : 0;
    if (!live_scan && part && need && need <= part_bytes && n_kv >= attn_split_min(v_trans) && (split_on || n_kv*4 > 48*1024)) {
        int64_t ns = (n_kv + 127)/128 < 32 ? (n_kv + 127)/128 : 32;       // ranges of >= 128 cells
        const int64_t chunk = ((n_kv + ns - 1)/ns + 31)/32*32;          // the KV cache pads n_kv to 32 (256 with flash attention)
        ns = (n_kv + chunk - 1)/chunk;
        a.kv_chunk = (int) chunk; a.nsplit = (int) ns; a.part = part;
    }
    const dim3 grid((unsigned) n_head, (unsigned) T, (unsigned) a.nsplit);
    const size_t lds = (((size_t)(a.nsplit > 1 ? a.kv_chunk : n_kv)*4 + 15) & ~(size_t) 15) + (v_trans ? 0 : 8192);     // !v_trans: + [256/(hd/8)][hd] partial sums
    if (!v_trans) {
#define MI_AD(KT_, VY_) do { if (head_dim == 128) hipLaunchKernelGGL((k_attn_decode<128, false, KT_, VY_>), grid, dim3(256), lds, stream, a); \
                             else                 hipLaunchKernelGGL((k_attn_decode<64, false, KT_, VY_>),  grid, dim3(256), lds, stream, a); } while (0)
        if (kt == T_F16 && vy == T_F16)        MI_AD(T_F16, T_F16);
        else if (kt == T_Q8_0 && vy == T_Q8_0) MI_AD(T_Q8_0, T_Q8_0);
        else if (kt == T_Q4_0 && vy == T_Q4_0) MI_AD(T_Q4_0, T_Q4_0);
        else if (kt == T_BF16 && vy == T_BF16) MI_AD(T_BF16, T_BF16);
        else if (kt == T_Q8_0 && vy == T_F16)  MI_AD(T_Q8_0, T_F16);
        else                                   MI_AD(T_Q8_0, T_Q4_0);
#undef MI_AD
    }
    if (v_trans && kt == T_Q8_0) {
        if (head_dim == 128) hipLaunchKernelGGL((k_attn_decode<128, true, T_Q8_0>), grid, dim3(256), lds, stream, a);
        else                 hipLaunchKernelGGL((k_attn_decode<64, true, T_Q8_0>),  grid, dim3(256), lds, stream, a);
    } else
    if (v_trans) {
        if (head_dim == 128) hipLaunchKernelGGL((k_attn_decode<128>), grid, dim3(256), lds, stream, a);
        else                 hipLaunchKernelGGL((k_attn_decode<64>),  grid, dim3(256), lds, stream, a);
    }
    if (a.nsplit > 1) {
        const dim3 mg((unsigned) n_head, (unsigned) T);
        if (head_dim == 128) hipLaunchKernelGGL((k_attn_merge<128>), mg, dim3(128), 0, stream, a);
        else                 hipLaunchKernelGGL((k_attn_merge<64>),  mg, dim3(64),  0, stream, a);
    }
}

} // namespace mi355x
