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
#include <atomic>
#include <mutex>
#include "mmvq_core.h"
#include "quant_core.h"
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
};

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
template <int HD, bool VT = true>
__global__ void __launch_bounds__(256) k_attn_decode(const attn_args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float * s = (float *) smem;                          // [n_kv] scores -> probabilities
    __shared__ float sh[4];
    const int h = blockIdx.x, t = blockIdx.y;
    const int hk = h/(p.n_head/p.n_head_kv);
    const bool split = p.nsplit > 1;
    const int kv_lo = split ? (int) blockIdx.z*p.kv_chunk : 0, kv_n = split ? min(p.n_kv - kv_lo, p.kv_chunk) : p.n_kv;   // this workgroup's cells
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPC = HD/8;                            // lanes per K row (8 f16 = 16 B each)
    constexpr int CPW = 64/LPC;                          // K rows per wave step
    constexpr int U = 4;
    const int sub = lane % LPC, cw = lane / LPC;

    // ---- scores: s[j] = scale * K[j].q + mask[j] ----
    const float * qp = (const float *) (p.q + (size_t) t*p.q_nb1 + (size_t) h*p.q_nb2) + sub*8;
    const float4v q0 = *(const float4v *) qp, q1 = *(const float4v *) (qp + 4);
    const char * kbase = p.k + (size_t) hk*p.k_nb2 + sub*16 + (size_t) kv_lo*p.k_nb1;
    const char * mrow = p.mask ? p.mask + (size_t) t*p.m_nb1 + (size_t) kv_lo*(p.mask_f16 ? 2 : 4) : nullptr;
    float mx = (p.sinks && !split) ? p.sinks[h] : -INFINITY;
    for (int j0 = wave*CPW + cw; j0 < kv_n; j0 += 4*CPW*U) {
        int4v kreg[U]; float mreg[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = min(j0 + u*4*CPW, kv_n - 1);
            kreg[u] = *(const int4v *) (kbase + (size_t) j*p.k_nb1);
            mreg[u] = 0.0f;
            if (mrow) mreg[u] = p.mask_f16 ? f16_bits_to_f32(*(const uint16_t *) (mrow + (size_t) j*2)) : *(const float *) (mrow + (size_t) j*4);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = j0 + u*4*CPW;
            float acc = dot8_f16_f32(kreg[u], q0, q1);
            acc += dpp_f<0xB1>(acc); acc += dpp_f<0x4E>(acc); acc += dpp_f<0x141>(acc);   // sum over the LPC lanes of the row
            if (LPC == 16) acc += dpp_f<0x140>(acc);
            if (j < kv_n) {
                const float v = acc*p.scale + mreg[u];
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
    for (int j = threadIdx.x; j < kv_n; j += 256) s[j] *= inv;   // the unfused SOFT_MAX normalises before V.p
    __syncthreads();
    // where the result goes: the output row, or this range's slot of the partial buffer (+ its max and denominator)
    float * orow = split ? p.part + ((size_t)(t*p.n_head + h)*p.nsplit + blockIdx.z)*(HD + 2) : (float *) ((char *) p.dst + (size_t) t*p.dst_nb1) + (size_t) h*HD;
    if (split && threadIdx.x == 0) { orow[HD] = mx; orow[HD + 1] = sum; }

    if (!VT) {
        // ---- out[d] = sum_j p[j]*V[j][d]: a thread owns 8 dims of every NGR-th cell; partial sums meet in LDS ----
        constexpr int DCH = HD/8, NGR = 256/DCH, UV = 4;
        float * red = (float *) (smem + (((size_t)(split ? p.kv_chunk : p.n_kv)*4 + 15) & ~(size_t) 15));     // [NGR][HD]
        const int dch = threadIdx.x % DCH, cg = threadIdx.x / DCH;
        const char * vb = p.v + (size_t) hk*p.v_nb2 + (size_t) dch*16 + (size_t) kv_lo*p.v_nb1;
        float a8[8] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
        for (int j0 = cg; j0 < kv_n; j0 += NGR*UV) {
            int4v vr[UV]; float pj[UV];
#pragma unroll
            for (int u = 0; u < UV; u++) {
                const int j = min(j0 + u*NGR, kv_n - 1);
                vr[u] = *(const int4v *) (vb + (size_t) j*p.v_nb1);
                pj[u] = j0 + u*NGR < kv_n ? s[j] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < UV; u++) {
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
        for (int g = 0; g < NG; g++) vreg[g] = ld_b128(vbase + (size_t)(g*16)*p.v_nb1 + (size_t) c*16);
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

void attn_decode(const void * q, size_t q_nb1, size_t q_nb2, const void * k, size_t k_nb1, size_t k_nb2, const void * v, size_t v_nb1, size_t v_nb2,
                 const void * mask, size_t m_nb1, bool mask_f16, const float * sinks, float * dst, size_t dst_nb1,
                 int64_t head_dim, int64_t n_kv, int64_t n_head, int64_t n_head_kv, int64_t T, float scale, hipStream_t stream, bool v_trans,
                 float * part, size_t part_bytes) {
    attn_args a = { (const char *) q, q_nb1, q_nb2, (const char *) k, k_nb1, k_nb2, (const char *) v, v_nb1, v_nb2,
                    (const char *) mask, m_nb1, mask_f16 ? 1 : 0, sinks, dst, dst_nb1, (int) n_kv, (int) n_head, (int) n_head_kv, (int) T, scale, 0, 1, nullptr };
    // long contexts: one workgroup per (head, token) walks every cell alone — 32 workgroups on 256 CUs. With a partial buffer the cells
    // are split into ranges of >= 256 (at most 32 ranges) that run side by side and a small second kernel merges them.
    static const bool split_on = !getenv("GGML_MI355X_ATTN_SPLIT") || atoi(getenv("GGML_MI355X_ATTN_SPLIT")) != 0;
    const size_t need = attn_decode_part_bytes(head_dim, n_kv, n_head, T);
    if (part && need && need <= part_bytes && n_kv >= attn_split_min(v_trans) && (split_on || n_kv*4 > 48*1024)) {
        int64_t ns = (n_kv + 127)/128 < 32 ? (n_kv + 127)/128 : 32;       // ranges of >= 128 cells
        const int64_t chunk = ((n_kv + ns - 1)/ns + 31)/32*32;          // the KV cache pads n_kv to 32 (256 with flash attention)
        ns = (n_kv + chunk - 1)/chunk;
        a.kv_chunk = (int) chunk; a.nsplit = (int) ns; a.part = part;
    }
    const dim3 grid((unsigned) n_head, (unsigned) T, (unsigned) a.nsplit);
    const size_t lds = (((size_t)(a.nsplit > 1 ? a.kv_chunk : n_kv)*4 + 15) & ~(size_t) 15) + (v_trans ? 0 : 8192);     // !v_trans: + [256/(hd/8)][hd] partial sums
    if (!v_trans) {
        if (head_dim == 128) hipLaunchKernelGGL((k_attn_decode<128, false>), grid, dim3(256), lds, stream, a);
        else                 hipLaunchKernelGGL((k_attn_decode<64, false>),  grid, dim3(256), lds, stream, a);
    }
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

// ---------------------------------------------------------------------------------------------------------------
// grouped mat-vec with epilogues (n = 1)
// ---------------------------------------------------------------------------------------------------------------

struct fused_mmvq_args {
    mmvq_group g[MMVQ_MAX_GROUPS];
    int n_groups;
    int block_end[MMVQ_MAX_GROUPS];       // cumulative workgroup counts
    int k;
    int act_kind;
    // PRO_Q8: quantized activation column as ONE contiguous image in global memory (qs | d | bsums at the offsets
    // act_q8_carve gives for n = 1), staged verbatim into LDS. PRO_QUANT / PRO_NORM build the same image in LDS from x.
    const char * act; int act_chunks;     // 16-byte chunks
    int off_d, off_bs;                    // byte offsets of d / bsums inside the image
    const float * x; const float * norm_w; float eps;
    fused_rope rope;
#ifdef MI_STAMPS
    unsigned long long * stamps;          // [workgroup][8] 100 MHz wall-clock stamps (tools/stamp_timeline.py), NULL = off
#endif
};

#ifdef MI_STAMPS
#define MI_STAMP(i_) do { if (p.stamps && threadIdx.x == 0) p.stamps[blockIdx.x*8 + (i_)] = wall_clock64(); } while (0)
#else
#define MI_STAMP(i_) do { } while (0)
#endif


// PERSISTENT grouped mat-vec. A launch has ONE workgroup of 8 (or 16) waves per CU; each workgroup belongs to one group (weight
// tensor) and its waves walk that tensor's row pairs (single rows for the dual GLU stream) with a grid stride, so that
//   * the activation is prepared ONCE per workgroup (copy / quantize / rms-norm + quantize into LDS) instead of once per 8 rows,
//   * the packed-weight stream never stops: loads run D steps ahead across row boundaries in a STATIC ring of register sets (the
//     loop is unrolled D times; a rotating copy w0 = w1 makes the compiler wait for every outstanding load at the top of each step,
//     measured: tools/stamp_timeline.py), and the DPP reduction + epilogue of one row pair overlaps the loads of the next.
// Order of issue: (1) activation loads by every wave, workgroup barrier (a CU returns loads in request order: nothing HBM-bound
// may be queued in front of them), (2) norm weights, then the D steps of weight loads one at a time BETWEEN the phases of
// (3) the prologue into LDS (a wave that cannot queue a load cannot do its share of the prologue either) + barrier,
// (4) integer dots, (5) reduction + epilogue per row pair.
// Every load is unconditional (clamped address) so that the number of outstanding loads is the same on every path.
//   PRO  : where the activation comes from (mmvq_prologue)
//   NA   : PRO_Q8: 16-byte image chunks per thread; PRO_QUANT/PRO_NORM: 256-element chunks per wave (k <= NA*256*waves)
//   D    : ring depth (2; 4 for the one-row GLU units and for long single-tensor streams)
constexpr int FW = 8;            // waves per workgroup

// what lane 0 does with the two finished rows of a pair (inlined: a call would spill the in-flight prefetch registers).
// CHAIN: the rows are handed to the next phase of the same launch, which runs on other CUs behind other L2s: stores and loads of
// handed-over bytes bypass the caches that are not coherent (agent-scope relaxed atomics lower to `sc1` accesses).
struct pair_out { float s0, s1; int row0; int pos0; long long idx0; };
template <bool CHAIN>
static __device__ __forceinline__ float ld_handoff(const float * p) {
    return CHAIN ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool CHAIN>
static __device__ __forceinline__ void st_handoff(float * p, float v) {
    if (CHAIN) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool CHAIN>
static __device__ __forceinline__ void finish_pair(const mmvq_group & g, const fused_rope & rope, pair_out o, const int rows) {
    float s0 = o.s0, s1 = o.s1;
    const int row0 = o.row0;
    const int m = rows > 1 ? g.m : row0 + 1;      // rows == 1: the unit has no second row
    if (g.epi == EPI_ADD) {
        s0 += ld_handoff<CHAIN>(g.res + row0);
        if (row0 + 1 < m) s1 += ld_handoff<CHAIN>(g.res + row0 + 1);
    } else if (g.epi == EPI_ROPE) {
        rope_pair(rope, o.pos0, row0 % rope.head_dim, s0, s1);   // m is even on this path
    }
    st_handoff<CHAIN>(g.dst + row0, s0);
    if (row0 + 1 < m) st_handoff<CHAIN>(g.dst + row0 + 1, s1);
    if (g.st_mode == 1) {
        uint16_t * q = g.st16 + o.idx0*g.st_row_elems + row0;
        q[0] = f32_to_f16_bits(s0);
        if (row0 + 1 < m) q[1] = f32_to_f16_bits(s1);
    } else if (g.st_mode == 2) {
        g.st16[g.st_idx[row0]] = f32_to_f16_bits(s0);
        if (row0 + 1 < m) g.st16[g.st_idx[row0 + 1]] = f32_to_f16_bits(s1);
    }
}

// hand-off between two phases of one launch (k_mmvq_chain): phase n waits until `target` workgroups have added to `ctr`
struct chain_wait { unsigned * ctr; unsigned target; unsigned * err; };

template <int TYPE, bool GLU, int PRO, int NA, int D, bool CHAIN = false, int FWT = 8>
static __device__ __forceinline__ void fused_body(const mmvq_group & g, const fused_mmvq_args & p, char * smem, int wg_in_group, int nwg_group,
                                                  int lane, int wave, const chain_wait cw = chain_wait{ nullptr, 0, nullptr }) {
    typedef mmvq_t<TYPE> T;
    // rows per unit of work: a pair for single-tensor groups; ONE row (of gate and of up) for the dual GLU stream, so that n_ff = 14336
    // rows split evenly over 2048 waves (7 each; as pairs it was 4 for half the waves and 3 for the rest — tools/stamp_timeline.py)
    constexpr int R = GLU ? 1 : 2, LPB = T::LPB, BPW = 64/LPB, ACT = T::ACT;
    const int nb = p.k / T::QK;
    const int iters = (nb + BPW - 1)/BPW;
    const int slot = lane % LPB, ibl = lane / LPB;
    const int P = (g.m + R - 1)/R;                       // row pairs in this group
    const int stride = nwg_group*FWT;
    const int p_first = wg_in_group*FWT + wave;
    int p_cur = p_first;

    MI_STAMP(0);
    // an expert of a stack (MUL_MAT_ID, one token): the index is a device value, workgroup-uniform -> scalar load
    const size_t eoff = g.eid ? (size_t) g.eid[0]*g.estride : 0;
    const char * gW = g.W + eoff; const char * gW2 = GLU ? g.W2 + eoff : nullptr;
    const float * gx = p.x + g.x_off;
    int4v areg[PRO == PRO_Q8 ? NA : 1];
    float4v xv[PRO != PRO_Q8 ? NA : 1], wv[PRO == PRO_NORM ? NA : 1];
    const int nchunk = (p.k + 255) >> 8;     // the last chunk may be partial (k % 32 == 0 with Q8_0 activations: gpt-oss's 2880)
    // element offset of this lane's 4 floats in chunk slot i: clamped into the vector; `live` tells whether they exist
#define MI_XOFF(i_) min(min(wave + FWT*(i_), nchunk - 1)*256 + lane*4, p.k - 4)
#define MI_XLIVE(i_) ((wave + FWT*(i_))*256 + lane*4 < p.k)
    // the stream is the sequence of (row pair, k-step) this wave will consume; (p_pf, it_pf) is the next step to fetch.
    // Past the end of the stream the loads go to the wave's own first block (an L1 hit), not to a line every wave would share.
    int p_pf = p_cur, it_pf = 0;
    typename T::wfrag w[D][R], u[GLU ? D : 1][R];
#define MI_FETCH(d_) { \
        const bool live = p_pf < P; \
        const int pp = live ? p_pf : min(p_first, P - 1); \
        const int ibf = live ? min(it_pf*BPW + ibl, nb - 1) : 0; \
        _Pragma("unroll") for (int r = 0; r < R; r++) { \
            const size_t off = (size_t) min(pp*R + r, g.m - 1)*g.row_stride; \
            w[d_][r] = T::load_w(gW + off, ibf, slot); \
            if (GLU) u[GLU ? d_ : 0][r] = T::load_w(gW2 + off, ibf, slot); \
        } \
        if (++it_pf == iters) { it_pf = 0; p_pf += stride; } }
#define MI_FENCE asm volatile("" ::: "memory")
    // wave-uniform epilogue operands, fetched now (scalar loads) instead of at the end of the first row pair
    const int pos0 = g.epi == EPI_ROPE ? p.rope.pos[0] : 0;
    const long long idx0 = g.st_mode == 1 ? (long long) g.st_idx[0] : 0;

    if (CHAIN) {
        // A phase of k_mmvq_chain. What does not depend on the previous phase is requested BEFORE the hand-off wait: the norm
        // weights and the whole ring of weight steps — they arrive while the wait, the activation round trip and the prologue
        // arithmetic run, which between separate launches is time HBM spends idle.
        if (PRO == PRO_NORM) {
#pragma unroll
            for (int i = 0; i < NA; i++) wv[i] = *(const float4v *) (p.norm_w + MI_XOFF(i));
        }
#pragma unroll
        for (int d = 0; d < D; d++) MI_FETCH(d)
        MI_FENCE;
        if (cw.target && threadIdx.x == 0) {       // one relaxed sc1 poll loop per workgroup; bounded, so a lost hand-off cannot hang the GPU
            int spins = 0;
            while (__hip_atomic_load(cw.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < cw.target) {
                if (++spins > (1 << 23)) { __hip_atomic_store(cw.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __syncthreads();                           // every wave loads the handed-over bytes only behind the polling wave's barrier
        MI_FENCE;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const float * px = gx + MI_XOFF(i);
            xv[i].x = ld_handoff<true>(px); xv[i].y = ld_handoff<true>(px + 1); xv[i].z = ld_handoff<true>(px + 2); xv[i].w = ld_handoff<true>(px + 3);
            if (!MI_XLIVE(i)) xv[i] = float4v{ 0.0f, 0.0f, 0.0f, 0.0f };
        }
    } else {
    // ---- (1) activation loads ----
    if (PRO == PRO_Q8) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int idx = min((int) threadIdx.x + i*(FWT*64), p.act_chunks - 1);
            areg[i] = *(const int4v *) (p.act + (size_t) idx*16);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NA; i++) { xv[i] = *(const float4v *) (gx + MI_XOFF(i)); if (!MI_XLIVE(i)) xv[i] = float4v{ 0.0f, 0.0f, 0.0f, 0.0f }; }
    }
    // A CU's L1 returns data in request order across all its waves: a load that hits L2 (the activation, just written) queued
    // behind one that goes to HBM (weights, norm weights) of ANY wave comes back with HBM latency — 1-4 us instead of ~0.5 us,
    // and the whole prologue hangs on it (measured, tools/stamp_timeline.py: the second workgroup on a CU saw its activation 2 us
    // after the first). So: every wave issues its activation loads, the workgroup meets at a barrier (issue order = request
    // order), and only then are norm weights and the weight stream requested. The asm statements are compiler barriers too.
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (PRO == PRO_NORM) {
#pragma unroll
        for (int i = 0; i < NA; i++) wv[i] = *(const float4v *) (p.norm_w + min(wave + FWT*i, nchunk - 1)*256 + lane*4);
    }

    // ---- (2) weight prefetch: the first D steps of this wave's stream ----
    // A wave blocks at a load it cannot queue (the CU's request queue is finite) and then cannot run its share of the prologue
    // either, so the D steps are not issued in one burst: one step now, the others between the phases of the prologue (FENCE keeps
    // the compiler from hoisting them back up). HBM then has work from the first 0.2 us on and the prologue math starts as soon as
    // the activation is there.
    MI_FETCH(0)
    MI_FENCE;
    }

    // ---- (3) prologue: build the quantized activation image in LDS ----
    int8_t * l_qs = (int8_t *) smem; float * l_d = (float *) (smem + p.off_d); int16_t * l_bs = (int16_t *) (smem + p.off_bs);
    if (PRO == PRO_Q8) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int idx = threadIdx.x + i*(FWT*64);
            if (idx < p.act_chunks) *(int4v *) (smem + (size_t) idx*16) = areg[i];
        }
    } else {
        float scale = 1.0f;
        if (PRO == PRO_NORM) {
            float * red = (float *) (smem + p.off_bs + (((p.k >> (ACT == T_Q8_0 ? 5 : 4))*2 + 15) & ~15));   // FWT floats after the image
            float ss = 0.0f;
#pragma unroll
            for (int i = 0; i < NA; i++) if (wave + FWT*i < nchunk) ss += xv[i].x*xv[i].x + xv[i].y*xv[i].y + xv[i].z*xv[i].z + xv[i].w*xv[i].w;
            ss = wave_sum(ss);
            MI_STAMP(4);
            if (lane == 0) red[wave] = ss;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            ss = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
            if (FWT == 16) ss += ((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15]));
            scale = 1.0f/sqrtf(ss/(float) p.k + p.eps);
            MI_STAMP(7);
            MI_FENCE;
            if (!CHAIN) { MI_FETCH(1) }
            MI_FENCE;
        }
        // all chunks of this wave are quantized first (independent chains the scheduler can interleave; a chunk past the end is
        // quantized too — its lanes hold a clamped duplicate — and simply not stored), then stored
        uint32_t qp[NA]; float qd[NA]; int qb[NA];
#pragma unroll
        for (int i = 0; i < NA; i++) {
            float4v v = xv[i];
            if (PRO == PRO_NORM) { v.x = (v.x*scale)*wv[i].x; v.y = (v.y*scale)*wv[i].y; v.z = (v.z*scale)*wv[i].z; v.w = (v.w*scale)*wv[i].w; }
            qp[i] = quant_chunk256<ACT>(v, qd[i], qb[i]);
        }
        MI_STAMP(5);
        MI_FENCE;
        if (!CHAIN) {     // the steps must be fetched in ring order: set d holds stream step d
            if (PRO == PRO_QUANT) { MI_FETCH(1) }
            else if (D > 2)       { MI_FETCH(2) }
        }
        MI_FENCE;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int c = wave + FWT*i;
            if (c < nchunk) store_chunk256<ACT>(qp[i], qd[i], qb[i], c, lane, l_qs, l_d, l_bs);
        }
    }
    MI_FENCE;
    if (!CHAIN) {
        if (PRO == PRO_Q8) { MI_FETCH(1) }
        if (D > 2 && PRO != PRO_NORM) { MI_FETCH(2) }
        if (D > 3) { MI_FETCH(3) }
    }
    MI_FENCE;
    MI_STAMP(6);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    act_view av;
    av.qs = l_qs; av.d = l_d; av.bs = l_bs;
    MI_STAMP(1);

    // ---- (4)+(5) stream: D steps per trip, each consuming one register set and refilling it for D steps later ----
    const int my_pairs = p_cur < P ? (P - 1 - p_cur)/stride + 1 : 0;
    const int total = my_pairs*iters;
    int it = 0;
    float acc[2] = { 0.0f, 0.0f }, acu[2] = { 0.0f, 0.0f };
#ifdef MI_STAMPS
    bool first_pair = true;
#endif
    for (int s = 0; s < total; s += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (s + d < total) {        // wave-uniform
                const int ib = it*BPW + ibl;
                if (ib < nb) {
                    const typename T::afrag a = T::load_a(av, ib, slot);
#pragma unroll
                    for (int r = 0; r < R; r++) { acc[r] += T::dot(w[d][r], a, slot); if (GLU) acu[r] += T::dot(u[GLU ? d : 0][r], a, slot); }
                }
                MI_FETCH(d)
                if (++it == iters) {
#ifdef MI_STAMPS
                    if (first_pair) { MI_STAMP(2); first_pair = false; }
#endif
                    float s0 = wave_sum(acc[0]), s1 = R > 1 ? wave_sum(acc[1]) : 0.0f;
                    if (GLU) {
                        float up_s = wave_sum(acu[0]);
                        if (g.b_gate) {     // + bias rows of this group's expert (ADD_ID), wave-uniform addresses
                            const size_t brow = (size_t)(g.eid ? g.eid[0] : 0)*g.m + p_cur;
                            s0 += g.b_gate[brow]; up_s += g.b_up[brow];
                        }
                        if (g.glu_alpha != 0.0f) {      // swiglu_oai, as elem.hip k_glu
                            const float xc = fminf(s0, g.glu_limit), gc = fmaxf(fminf(up_s, g.glu_limit), -g.glu_limit);
                            s0 = (xc/(1.0f + expf(-xc*g.glu_alpha)))*(gc + 1.0f);
                        } else {
                            s0 = (s0/(1.0f + expf(-s0)))*up_s;      // silu(gate)*up, as elem.hip k_glu
                        }
                    }
                    if (lane == 0) finish_pair<CHAIN>(g, p.rope, pair_out{ s0, s1, p_cur*R, pos0, idx0 }, R);
                    it = 0; p_cur += stride;
                    acc[0] = acc[1] = 0.0f; acu[0] = acu[1] = 0.0f;
                }
            }
        }
    }
    MI_STAMP(3);
#undef MI_FETCH
#undef MI_XOFF
#undef MI_XLIVE
}

// One instantiation per {weight type or pair of types} x {GLU} x {prologue} x {activation size class}: a single kernel switching
// over all six formats at run time allocates registers for the fattest path (227 VGPRs -> 2 waves/SIMD), which starves the HBM stream.
// FWT = waves per workgroup: 8, or 16 for launches with more row pairs than 8 waves x CUs but no more than 16 x CUs (norm + QKV:
// 3072 pairs) — ONE 1024-thread workgroup per CU shares one prologue (two 8-wave workgroups on a CU ran the second one's prologue
// ~2x slower), every wave owns a single pair, and the prologue has one 256-chunk per wave instead of two
template <int TA, int TB, bool GLU, int PRO, int NA, int D, int FWT = 8>
__global__ void __launch_bounds__(FWT*64, FWT == 8 ? 2 : 1) k_mmvq_fused(const fused_mmvq_args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int gi = 0;
    while (gi < p.n_groups - 1 && (int) blockIdx.x >= p.block_end[gi]) gi++;
    const int first = gi ? p.block_end[gi - 1] : 0;
    const int blk = (int) blockIdx.x - first, nwg = p.block_end[gi] - first;
    const mmvq_group & g = p.g[gi];
    if (TA == TB || g.type == TA) fused_body<TA, GLU, PRO, NA, D, false, FWT>(g, p, smem, blk, nwg, lane, wave);
    else                          fused_body<TB, GLU, PRO, NA, D, false, FWT>(g, p, smem, blk, nwg, lane, wave);
}

// ---- several grouped mat-vec launches of one decode layer as ONE launch ----
// Positions are fixed (the order llm_build_llama emits them, src/llama-model.cpp:6042-6100): 0 quantize + wo + residual,
// 1 norm + gate/up + SwiGLU, 2 quantize + down + residual, 3 norm + QKV + RoPE + KV store of the NEXT layer (or norm + lm_head);
// a launch runs the contiguous positions [first, last]. Between positions the rows are handed over through `sc1` stores / loads
// and one agent-scope counter per position (MI355X guide, "inter-workgroup visibility": every storing wave waits for its stores,
// the workgroup meets at a barrier, one lane adds to the counter; the consumer polls it with one lane, then a workgroup barrier,
// then `sc1` loads). What a separate launch pays for — dispatch, end-of-kernel, and HBM idling until the first weights arrive —
// is replaced by the counter round trip with the next position's weights already in flight.
// One workgroup per CU and the whole grid resident (grid <= CU count), or the wait could never be satisfied; the poll is bounded.
struct chain_args { int first, last; unsigned * sync; fused_mmvq_args ph[4]; };      // sync: [0..3] per-position counters, [6] finished, [7] error

template <int TA, int TB>
__global__ void __launch_bounds__(512, 1) k_mmvq_chain(const chain_args c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned prev_total = 0;
#define MI_PHASE(POS_, GLU_, PRO_, NA_, D_) \
    if (c.first <= POS_ && POS_ <= c.last) { \
        const fused_mmvq_args & p = c.ph[POS_]; \
        const int total = p.block_end[p.n_groups - 1]; \
        const chain_wait cw = { c.sync + (POS_ > 0 ? POS_ - 1 : 0), POS_ > c.first ? prev_total : 0u, c.sync + 7 }; \
        if ((int) blockIdx.x < total) { \
            int gi = 0; \
            while (gi < p.n_groups - 1 && (int) blockIdx.x >= p.block_end[gi]) gi++; \
            const int first_b = gi ? p.block_end[gi - 1] : 0; \
            const mmvq_group & g = p.g[gi]; \
            if (TA == TB || g.type == TA) fused_body<TA, GLU_, PRO_, NA_, D_, true>(g, p, smem, (int) blockIdx.x - first_b, p.block_end[gi] - first_b, lane, wave, cw); \
            else                          fused_body<TB, GLU_, PRO_, NA_, D_, true>(g, p, smem, (int) blockIdx.x - first_b, p.block_end[gi] - first_b, lane, wave, cw); \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* every storing wave: its stores have left */ \
            __syncthreads();                                       /* also: the LDS image is free for the next position */ \
            if (threadIdx.x == 0 && POS_ < c.last) __hip_atomic_fetch_add(c.sync + POS_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
        } \
        prev_total = (unsigned) total; \
    }
    MI_PHASE(0, false, PRO_QUANT, 2, 2)
    MI_PHASE(1, true,  PRO_NORM,  2, 4)
    MI_PHASE(2, false, PRO_QUANT, 8, 2)
    MI_PHASE(3, false, PRO_NORM,  2, 2)
#undef MI_PHASE
    // the last workgroup to finish re-arms the counters for the next launch that uses this slot (a graph replay)
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(c.sync + 6, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            for (int i = 0; i < 7; i++) __hip_atomic_store(c.sync + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#ifdef MI_STAMPS
// debug build only (-DMI_STAMPS): every grouped mat-vec launch gets the next slot of a device buffer; slots are baked into captured graphs
static unsigned long long * g_stamp_buf = nullptr;
static int g_stamp_slots = 0, g_stamp_next = 0;
constexpr int STAMP_MAX_WG = 1024;
struct stamp_meta { int blocks, k, rows, type_a, type_b, mode, glu; long long bytes; };
static stamp_meta g_stamp_meta[4096];
extern "C" int mi355x_stamps_enable(int n_slots) {
    if (g_stamp_buf) { (void) hipFree(g_stamp_buf); g_stamp_buf = nullptr; }
    g_stamp_slots = n_slots > 4096 ? 4096 : n_slots; g_stamp_next = 0;
    if (g_stamp_slots <= 0) return 0;
    if (hipMalloc(&g_stamp_buf, (size_t) g_stamp_slots*STAMP_MAX_WG*8*8) != hipSuccess) return -1;
    (void) hipMemset(g_stamp_buf, 0, (size_t) g_stamp_slots*STAMP_MAX_WG*8*8);
    return 0;
}
extern "C" int mi355x_stamps_used(void) { return g_stamp_next; }
extern "C" int mi355x_stamps_read(int slot, unsigned long long * out, int * meta, long long * bytes) {
    if (!g_stamp_buf || slot < 0 || slot >= g_stamp_slots) return -1;
    const stamp_meta & m = g_stamp_meta[slot];
    (void) hipMemcpy(out, g_stamp_buf + (size_t) slot*STAMP_MAX_WG*8, (size_t) m.blocks*8*8, hipMemcpyDeviceToHost);
    meta[0] = m.blocks; meta[1] = m.k; meta[2] = m.rows; meta[3] = m.type_a; meta[4] = m.type_b; meta[5] = m.mode; meta[6] = m.glu;
    *bytes = m.bytes;
    return 0;
}
#endif


static size_t pad256h(size_t x) { return (x + 255) & ~(size_t) 255; }

static size_t act_image_bytes(int64_t k, int act_kind) {
    const int64_t nd = act_kind == T_Q8_0 ? k/32 : k/256, nbs = act_kind == T_Q8_0 ? k/32 : k/16;
    return pad256h(k) + pad256h(nd*4) + ((nbs*2 + 15) & ~15);
}

// the activation must be the n = 1 image act_q8_carve lays out: qs | pad | d | pad | bsums, contiguous
bool mul_mat_vec_q_fused_supported(int64_t k, int act_kind) {
    return k % (act_kind == T_Q8_0 ? 32 : 256) == 0 && act_image_bytes(k, act_kind) <= 4*512*16;
}
bool mul_mat_vec_q_fused_prologue_supported(int64_t k, int act_kind) { return (k % 256 == 0 || (act_kind == T_Q8_0 && k % 32 == 0)) && k <= 16*1024; }

// per host thread: one backend (stream) is driven by one thread at a time, different backends concurrently from different threads
// (tests/test-thread-safety.cpp)
static thread_local struct { mmvq_launch_hook pre = nullptr, post = nullptr; void * ctx = nullptr; } g_hook;
void mul_mat_vec_q_fused_set_hooks(mmvq_launch_hook pre, mmvq_launch_hook post, void * ctx) { g_hook.pre = pre; g_hook.post = post; g_hook.ctx = ctx; }

// a grouped launch, prepared: either run at once or held back as a position of a chained launch
struct fused_launch { fused_mmvq_args a; int blocks; size_t lds; int ta, tb; bool glu; int mode, na; bool deep; int64_t k; uint64_t wbytes; int fw; };
static void fused_launch_now(const fused_launch & L, hipStream_t stream);

static fused_launch fused_prepare(const mmvq_group * groups, int n_groups, int64_t k, const mmvq_input & in, const mmvq_rope * rope) {
    fused_launch L = {};
    fused_mmvq_args & a = L.a;
    a.n_groups = n_groups; a.k = (int) k; a.act_kind = in.act_kind;
    // share the persistent workgroups among the groups in proportion to their rows (never more than one row pair per wave)
    static int n_cu = 0, wpc = 1, glu_wpc = 1;   // measured (tools/stamp_timeline.py): the second workgroup on a CU runs its prologue ~2x slower
    if (n_cu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
        if (const char * e = getenv("GGML_MI355X_MMVQ_WPC")) wpc = atoi(e) > 0 ? atoi(e) : 1;
        if (const char * e = getenv("GGML_MI355X_GLU_WPC")) glu_wpc = atoi(e) > 0 ? atoi(e) : 1;
    }
    int64_t rows_total = 0;
    for (int i = 0; i < n_groups; i++) rows_total += (int64_t) groups[i].m*(groups[i].epi == EPI_GLU ? 2 : 1);
    // 16 waves per workgroup: more than one row pair per wave at 8 waves x CUs, at most one at 16 (the in-prologue-norm launches only)
    static int fw16_env = -1;
    if (fw16_env < 0) { const char * e = getenv("GGML_MI355X_MMVQ_FW16"); fw16_env = e ? atoi(e) : 1; }
    const int FW = (fw16_env && groups[0].epi != EPI_GLU && in.mode == PRO_NORM && k <= 4096 && k % 256 == 0 &&
                    (rows_total + 1)/2 > (int64_t) n_cu*8 &&
                    ((rows_total + 1)/2 <= (int64_t) n_cu*16 + 64 || (rows_total + 1)/2 >= (int64_t) n_cu*64)) ? 16 : 8;   // or a long stream (lm_head: 101 -> 96 us)
    L.fw = FW;
    const int budget = n_cu*(groups[0].epi == EPI_GLU ? glu_wpc : wpc);   // the dual (GLU) kernels need > 128 VGPRs: one workgroup per CU
    int blocks = 0;
    for (int i = 0; i < n_groups; i++) {
        a.g[i] = groups[i];
        const int max_wg = groups[i].epi == EPI_GLU ? (int)((groups[i].m + FW - 1)/FW) : (int)(((groups[i].m + 1)/2 + FW - 1)/FW);   // units: rows (GLU) or row pairs
        int share = (int)(((int64_t) budget*groups[i].m*(groups[i].epi == EPI_GLU ? 2 : 1))/rows_total);   // rounded down: the grid never exceeds the budget (one workgroup per CU)
        share = share < 1 ? 1 : (share > max_wg ? max_wg : share);
        blocks += share;
        a.block_end[i] = blocks;
    }
    const int64_t nd = in.act_kind == T_Q8_0 ? k/32 : k/256;
    a.off_d = (int) pad256h(k);
    a.off_bs = (int)(pad256h(k) + pad256h(nd*4));
    const size_t bytes = act_image_bytes(k, in.act_kind);
    a.act_chunks = (int)(bytes/16);
    if (in.mode == PRO_Q8) {
        a.act = (const char *) in.act.qs;
        if ((const char *) in.act.d - (const char *) in.act.qs != a.off_d || (const char *) in.act.bsums - (const char *) in.act.qs != a.off_bs) {
            fprintf(stderr, "mul_mat_vec_q_fused: activation is not a contiguous n = 1 image\n"); abort();
        }
    } else {
        a.x = in.x; a.norm_w = in.norm_w; a.eps = in.eps;
    }
    if (rope) {
        a.rope = make_fused_rope(*rope);
    }
    const size_t lds = bytes + 64;     // + FW floats for the RMS reduction
    int ta = groups[0].type, tb = groups[0].type;
    for (int i = 1; i < n_groups; i++) if (groups[i].type != ta) tb = groups[i].type;
    if (tb < ta) { const int t = ta; ta = tb; tb = t; }
    const bool glu = groups[0].epi == EPI_GLU;
    const int mode = in.mode;
#ifdef MI_STAMPS
    a.stamps = nullptr;
    if (g_stamp_buf && g_stamp_next < g_stamp_slots && blocks <= STAMP_MAX_WG) {
        stamp_meta & sm = g_stamp_meta[g_stamp_next];
        sm.blocks = blocks; sm.k = (int) k; sm.rows = (int) rows_total; sm.type_a = ta; sm.type_b = tb; sm.mode = mode; sm.glu = glu;
        sm.bytes = 0;
        for (int i = 0; i < n_groups; i++) sm.bytes += (long long) groups[i].m*groups[i].row_stride*(groups[i].epi == EPI_GLU ? 2 : 1);
        a.stamps = g_stamp_buf + (size_t) g_stamp_next*STAMP_MAX_WG*8;
        g_stamp_next++;
    }
#endif
    const int na = mode == PRO_Q8 ? (a.act_chunks <= 512 ? 1 : (a.act_chunks <= 1024 ? 2 : 4)) : (FW == 16 ? 1 : (k <= 4096 ? 2 : 8));
    // prefetch depth: measured on Llama-3-8B Q4_K_M tg128 (profiles/r01_g_*): D = 2 everywhere 503 tok/s, D = 4 (3 for the dual GLU
    // stream) everywhere 482-486 — a CU's request queue is finite and a wave that cannot queue a load cannot run its share of the
    // prologue either. Only long single-tensor streams (>= 16 steps per wave: the lm_head, 31 row pairs per wave) take the deep ring;
    // GGML_MI355X_MMVQ_DEPTH=2|4 forces one for experiments (the GLU kernels exist with D = 2 only).
    static int depth_env = -1;
    if (depth_env < 0) { const char * e = getenv("GGML_MI355X_MMVQ_DEPTH"); depth_env = e ? atoi(e) : 0; }
    int64_t max_steps = 0;
    for (int i = 0; i < n_groups; i++) {
        const int nwg_i = a.block_end[i] - (i ? a.block_end[i - 1] : 0);
        const int64_t pairs = (groups[i].m + 1)/2, per_wave = (pairs + (int64_t) nwg_i*FW - 1)/((int64_t) nwg_i*FW);
        const int64_t nblk = k/(in.act_kind == T_Q8_0 ? 32 : 256);
        const int64_t it = in.act_kind == T_Q8_0 ? (nblk + 63)/64 : (nblk + 7)/8;      // <= the steps per row pair of every type
        if (per_wave*it > max_steps) max_steps = per_wave*it;
    }
    const bool deep = depth_env ? depth_env > 2 : max_steps >= 16;
    L.blocks = blocks; L.lds = lds; L.ta = ta; L.tb = tb; L.glu = glu; L.mode = mode; L.na = na; L.deep = deep; L.k = k;
    for (int i = 0; i < n_groups; i++) L.wbytes += (uint64_t) groups[i].m*groups[i].row_stride*(groups[i].epi == EPI_GLU ? 2 : 1);
    return L;
}

static void fused_launch_kernel(const fused_launch & L, hipStream_t stream);
static void fused_launch_now(const fused_launch & L, hipStream_t stream) {
    if (g_hook.pre) g_hook.pre(g_hook.ctx, L.ta, L.wbytes, 1, L.k);
    fused_launch_kernel(L, stream);
    if (g_hook.post) g_hook.post(g_hook.ctx, L.ta, L.wbytes, 1, L.k);
}
static void fused_launch_kernel(const fused_launch & L, hipStream_t stream) {
    const fused_mmvq_args & a = L.a;
    const dim3 grid((unsigned) L.blocks);
    const size_t lds = L.lds;
    const int ta = L.ta, tb = L.tb, mode = L.mode, na = L.na;
    const bool glu = L.glu, deep = L.deep;
    constexpr int FW = 8;
#define MI_L(TA_, TB_, GLU_, PRO_, NA_) do { \
        if (GLU_)      hipLaunchKernelGGL((k_mmvq_fused<TA_, TB_, true,  PRO_, NA_, 4>), grid, dim3(FW*64), lds, stream, a);   /* one-row units: 4 steps = the bytes 2 steps of pairs held */ \
        else if (deep) hipLaunchKernelGGL((k_mmvq_fused<TA_, TB_, false, PRO_, NA_, 4>), grid, dim3(FW*64), lds, stream, a); \
        else           hipLaunchKernelGGL((k_mmvq_fused<TA_, TB_, false, PRO_, NA_, 2>), grid, dim3(FW*64), lds, stream, a); } while (0)
 #define MI_LAUNCH(TA_, TB_, GLU_) do { \
        if (L.fw == 16) { hipLaunchKernelGGL((k_mmvq_fused<TA_, TB_, false, PRO_NORM, 1, 2, 16>), grid, dim3(1024), lds, stream, a); } \
        else if (mode == PRO_Q8)         { if (na == 1) MI_L(TA_, TB_, GLU_, PRO_Q8, 1); else if (na == 2) MI_L(TA_, TB_, GLU_, PRO_Q8, 2); else MI_L(TA_, TB_, GLU_, PRO_Q8, 4); } \
        else if (mode == PRO_NORM)  { if (na == 2) MI_L(TA_, TB_, GLU_, PRO_NORM, 2); else MI_L(TA_, TB_, GLU_, PRO_NORM, 8); } \
        else                        { if (na == 2) MI_L(TA_, TB_, GLU_, PRO_QUANT, 2); else MI_L(TA_, TB_, GLU_, PRO_QUANT, 8); } } while (0)
#define MI_SINGLE(T_) if (ta == T_ && tb == T_) { if (glu) MI_LAUNCH(T_, T_, true); else MI_LAUNCH(T_, T_, false); return; }
    MI_SINGLE(T_Q4_K) MI_SINGLE(T_Q6_K) MI_SINGLE(T_Q5_K) MI_SINGLE(T_Q8_0) MI_SINGLE(T_Q4_0) MI_SINGLE(T_MXFP4)
    if (ta == T_Q4_K && tb == T_Q5_K) { MI_LAUNCH(T_Q4_K, T_Q5_K, false); return; }
    if (ta == T_Q4_K && tb == T_Q6_K) { MI_LAUNCH(T_Q4_K, T_Q6_K, false); return; }
    if (ta == T_Q5_K && tb == T_Q6_K) { MI_LAUNCH(T_Q5_K, T_Q6_K, false); return; }
#undef MI_SINGLE
#undef MI_LAUNCH
#undef MI_L
    fprintf(stderr, "mul_mat_vec_q_fused: type pair (%d, %d) has no kernel (check mul_mat_vec_q_fused_can_group)\n", ta, tb);
    abort();
}

// ---- the chain queue: grouped launches that follow each other in a decode layer are held back and sent as ONE k_mmvq_chain ----
// (every caller that puts anything else on the stream calls mul_mat_vec_q_fused_flush first; backend.cpp does so per graph node)
static int chain_position(const fused_launch & L) {      // which fixed position of k_mmvq_chain this launch fits, or -1
    const bool types_ok = (L.ta == T_Q4_K || L.ta == T_Q6_K) && (L.tb == T_Q4_K || L.tb == T_Q6_K);
    if (!types_ok || L.deep || L.fw != 8) return -1;
    if (!L.glu && L.mode == PRO_QUANT && L.na == 2) return 0;
    if ( L.glu && L.mode == PRO_NORM  && L.na == 2) return 1;
    if (!L.glu && L.mode == PRO_QUANT && L.na == 8) return 2;
    if (!L.glu && L.mode == PRO_NORM  && L.na == 2) return 3;
    return -1;
}
static thread_local struct { fused_launch q[4]; int pos[4]; int n = 0; } g_chain_q;      // the held-back launches of this thread's stream
static struct { unsigned * sync = nullptr; std::atomic<int> next_slot{0}; int enabled = -1; int n_cu = 0; std::mutex init; } g_chain;
constexpr int CHAIN_SLOTS = 8192;

int mul_mat_vec_q_fused_pending(uint64_t * wbytes) {
    uint64_t b = 0;
    for (int i = 0; i < g_chain_q.n; i++) b += g_chain_q.q[i].wbytes;
    if (wbytes) *wbytes = b;
    return g_chain_q.n;
}

void mul_mat_vec_q_fused_flush(hipStream_t stream) {
    if (g_chain_q.n == 0) return;
    if (g_chain_q.n == 1) { g_chain_q.n = 0; fused_launch_now(g_chain_q.q[0], stream); return; }
    chain_args c = {};
    c.first = g_chain_q.pos[0]; c.last = g_chain_q.pos[g_chain_q.n - 1];
    size_t lds = 0; int blocks = 0;
    for (int i = 0; i < g_chain_q.n; i++) {
        c.ph[g_chain_q.pos[i]] = g_chain_q.q[i].a;
        if (g_chain_q.q[i].lds > lds) lds = g_chain_q.q[i].lds;
        if (g_chain_q.q[i].blocks > blocks) blocks = g_chain_q.q[i].blocks;
    }
    c.sync = g_chain.sync + (size_t)(g_chain.next_slot.fetch_add(1) % CHAIN_SLOTS)*8;
    uint64_t wb = 0;
    for (int i = 0; i < g_chain_q.n; i++) wb += g_chain_q.q[i].wbytes;
    const int n_merged = g_chain_q.n;
    g_chain_q.n = 0;
    if (g_hook.pre) g_hook.pre(g_hook.ctx, T_Q4_K, wb, n_merged, 0);
    hipLaunchKernelGGL((k_mmvq_chain<T_Q4_K, T_Q6_K>), dim3((unsigned) blocks), dim3(FW*64), lds, stream, c);
    if (g_hook.post) g_hook.post(g_hook.ctx, T_Q4_K, wb, n_merged, 0);
}

void mul_mat_vec_q_fused(const mmvq_group * groups, int n_groups, int64_t k, const mmvq_input & in, const mmvq_rope * rope, hipStream_t stream) {
    const fused_launch L = fused_prepare(groups, n_groups, k, in, rope);
    if (g_chain.enabled < 0) {
        std::lock_guard<std::mutex> lock(g_chain.init);
        if (g_chain.enabled < 0) {
            // opt-in: measured 446 tok/s chained vs 512 unchained (Llama-3-8B Q4_K_M tg128) — the sc1 hand-off (per-row 4-byte sc1 stores,
            // their acks before the counter add, sc1 reloads) costs more than the kernel boundary + weight wait it removes;
            // parity-tested, kept as the base for round 2
            const char * e = getenv("GGML_MI355X_CHAIN");
            int enabled = e ? atoi(e) : 0;
            int dev = 0; hipDeviceProp_t prop;
            g_chain.n_cu = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
            if (enabled) {      // counters: zeroed once; every chained launch re-arms its own slot when its last workgroup finishes
                if (hipMalloc(&g_chain.sync, (size_t) CHAIN_SLOTS*8*sizeof(unsigned)) != hipSuccess ||
                    hipMemset(g_chain.sync, 0, (size_t) CHAIN_SLOTS*8*sizeof(unsigned)) != hipSuccess) { (void) hipGetLastError(); enabled = 0; }
            }
            g_chain.enabled = enabled;
        }
    }
    int pos = g_chain.enabled ? chain_position(L) : -1;
#ifdef MI_STAMPS
    pos = -1;                                        // the timeline tool looks at separate launches
#endif
    if (pos >= 0 && (L.blocks > g_chain.n_cu || L.blocks > 1024)) pos = -1;      // the whole grid must be resident: one workgroup per CU
    if (pos < 0) {
        mul_mat_vec_q_fused_flush(stream);
        fused_launch_now(L, stream);
        return;
    }
    if (g_chain_q.n > 0 && pos != g_chain_q.pos[g_chain_q.n - 1] + 1) mul_mat_vec_q_fused_flush(stream);     // not the next position: a new chain starts here
    g_chain_q.q[g_chain_q.n] = L; g_chain_q.pos[g_chain_q.n] = pos; g_chain_q.n++;
    if (pos == 3) mul_mat_vec_q_fused_flush(stream);                                                    // the last position closes the chain
}

// which weight types may share one grouped launch (the mixtures llama_tensor_get_type produces, src/llama-quant.cpp:178-434)
bool mul_mat_vec_q_fused_can_group(int type_a, int type_b) {
    if (type_a == type_b) return true;
    const int lo = type_a < type_b ? type_a : type_b, hi = type_a < type_b ? type_b : type_a;
    return (lo == T_Q4_K && (hi == T_Q5_K || hi == T_Q6_K)) || (lo == T_Q5_K && hi == T_Q6_K);
}

} // namespace mi355x
