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
template <int ACT>
__global__ void __launch_bounds__(256) k_rms_norm_mul_quant(const float * __restrict__ x, size_t x_stride, const float * __restrict__ w,
                                                            float * __restrict__ y, size_t y_stride, int8_t * qs, float * d, int16_t * bs,
                                                            int ne0, float eps) {
    __shared__ float sh[4];
    const int row = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float * xr = (const float *) ((const char *) x + (size_t) row*x_stride);
    float * yr = (float *) ((char *) y + (size_t) row*y_stride);
    const int nchunk = ne0/256;
    float ss = 0.0f;
    for (int c = wave; c < nchunk; c += 4) {
        const float4v v = *(const float4v *) (xr + c*256 + lane*4);
        ss += v.x*v.x + v.y*v.y + v.z*v.z + v.w*v.w;
    }
    ss = block_sum4(ss, sh);
    const float scale = 1.0f/sqrtf(ss/(float) ne0 + eps);
    constexpr int ND = ACT == T_Q8_0 ? 32 : 256, NBS = ACT == T_Q8_0 ? 32 : 16;
    int8_t * qr = qs + (size_t) row*ne0; float * dr = d + (size_t) row*(ne0/ND); int16_t * br = bs + (size_t) row*(ne0/NBS);
    for (int c = wave; c < nchunk; c += 4) {
        float4v v = *(const float4v *) (xr + c*256 + lane*4);
        const float4v ww = *(const float4v *) (w + c*256 + lane*4);
        v.x = (v.x*scale)*ww.x; v.y = (v.y*scale)*ww.y; v.z = (v.z*scale)*ww.z; v.w = (v.w*scale)*ww.w;   // RMS_NORM then MUL: two roundings, as unfused
        *(float4v *) (yr + c*256 + lane*4) = v;
        quant_store_chunk256<ACT>(v, c, lane, qr, dr, br);
    }
}

void rms_norm_mul_quant(const float * x, size_t x_stride, const float * w, float * y, size_t y_stride, const act_q8 & q,
                        int64_t ne0, int64_t nrows, float eps, hipStream_t stream) {
    if (q.kind == T_Q8_0) hipLaunchKernelGGL((k_rms_norm_mul_quant<T_Q8_0>), dim3((unsigned) nrows), dim3(256), 0, stream, x, x_stride, w, y, y_stride, q.qs, q.d, q.bsums, (int) ne0, eps);
    else                  hipLaunchKernelGGL((k_rms_norm_mul_quant<T_Q8_K>), dim3((unsigned) nrows), dim3(256), 0, stream, x, x_stride, w, y, y_stride, q.qs, q.d, q.bsums, (int) ne0, eps);
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
};

template <int HD>
__global__ void __launch_bounds__(256) k_attn_decode(const attn_args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float * s = (float *) smem;                          // [n_kv] scores -> probabilities
    __shared__ float sh[4];
    const int h = blockIdx.x, t = blockIdx.y;
    const int hk = h/(p.n_head/p.n_head_kv);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPC = HD/8;                            // lanes per cell (8 f16 = 16 B each)
    constexpr int CPW = 64/LPC;                          // cells per wave step
    const int sub = lane % LPC, cw = lane / LPC;

    // this lane's 8 query elements
    const float * qp = (const float *) (p.q + (size_t) t*p.q_nb1 + (size_t) h*p.q_nb2) + sub*8;
    const float4v q0 = *(const float4v *) qp, q1 = *(const float4v *) (qp + 4);
    const char * kbase = p.k + (size_t) hk*p.k_nb2 + sub*16;
    const char * mrow = p.mask ? p.mask + (size_t) t*p.m_nb1 : nullptr;

    float mx = p.sinks ? p.sinks[h] : -INFINITY;
    for (int j0 = wave*CPW; j0 < p.n_kv; j0 += 4*CPW) {
        const int j = j0 + cw;
        float acc = 0.0f;
        if (j < p.n_kv) {
            const int4v kv = *(const int4v *) (kbase + (size_t) j*p.k_nb1);
            const uint32_t k0 = (uint32_t) kv.x, k1 = (uint32_t) kv.y, k2 = (uint32_t) kv.z, k3 = (uint32_t) kv.w;
            acc  = f16_bits_to_f32((uint16_t) k0)*q0.x + f16_bits_to_f32((uint16_t)(k0 >> 16))*q0.y;
            acc += f16_bits_to_f32((uint16_t) k1)*q0.z + f16_bits_to_f32((uint16_t)(k1 >> 16))*q0.w;
            acc += f16_bits_to_f32((uint16_t) k2)*q1.x + f16_bits_to_f32((uint16_t)(k2 >> 16))*q1.y;
            acc += f16_bits_to_f32((uint16_t) k3)*q1.z + f16_bits_to_f32((uint16_t)(k3 >> 16))*q1.w;
        }
        // sum over the LPC lanes of the cell (LPC = 16: a DPP row; LPC = 8: half a row)
        acc += dpp_f<0xB1>(acc); acc += dpp_f<0x4E>(acc); acc += dpp_f<0x141>(acc);
        if (LPC == 16) acc += dpp_f<0x140>(acc);
        if (j < p.n_kv) {
            float v = acc*p.scale;
            if (mrow) v += p.mask_f16 ? f16_bits_to_f32(*(const uint16_t *) (mrow + (size_t) j*2)) : *(const float *) (mrow + (size_t) j*4);
            if (sub == 0) s[j] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) sh[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    float sum = 0.0f;
    for (int j = threadIdx.x; j < p.n_kv; j += 256) { const float e = expf(s[j] - mx); s[j] = e; sum += e; }
    sum = block_sum4(sum, sh);
    if (p.sinks) sum += expf(p.sinks[h] - mx);
    const float inv = 1.0f/sum;
    for (int j = threadIdx.x; j < p.n_kv; j += 256) s[j] *= inv;   // the unfused SOFT_MAX normalises before V.p
    __syncthreads();

    // out[d] = sum_j V[d][j]*p[j]: one wave per output row d, lanes stride the cells 8 at a time
    const char * vbase = p.v + (size_t) hk*p.v_nb2;
    for (int d = wave; d < HD; d += 4) {
        const char * vr = vbase + (size_t) d*p.v_nb1;
        float acc = 0.0f;
        for (int j = lane*8; j < p.n_kv; j += 512) {
            if (j + 8 <= p.n_kv) {
                const int4v vv = ld_b128(vr + (size_t) j*2);
                const uint32_t v0 = (uint32_t) vv.x, v1 = (uint32_t) vv.y, v2 = (uint32_t) vv.z, v3 = (uint32_t) vv.w;
                acc += f16_bits_to_f32((uint16_t) v0)*s[j]     + f16_bits_to_f32((uint16_t)(v0 >> 16))*s[j + 1];
                acc += f16_bits_to_f32((uint16_t) v1)*s[j + 2] + f16_bits_to_f32((uint16_t)(v1 >> 16))*s[j + 3];
                acc += f16_bits_to_f32((uint16_t) v2)*s[j + 4] + f16_bits_to_f32((uint16_t)(v2 >> 16))*s[j + 5];
                acc += f16_bits_to_f32((uint16_t) v3)*s[j + 6] + f16_bits_to_f32((uint16_t)(v3 >> 16))*s[j + 7];
            } else {
                for (int jj = j; jj < p.n_kv; jj++) acc += f16_bits_to_f32(*(const uint16_t *) (vr + (size_t) jj*2))*s[jj];
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) *(float *) ((char *) p.dst + (size_t) t*p.dst_nb1 + (size_t)(h*HD + d)*4) = acc;
    }
}

bool attn_decode_supported(int64_t head_dim, int64_t n_kv) { return (head_dim == 128 || head_dim == 64) && n_kv*4 <= 60*1024; }

void attn_decode(const void * q, size_t q_nb1, size_t q_nb2, const void * k, size_t k_nb1, size_t k_nb2, const void * v, size_t v_nb1, size_t v_nb2,
                 const void * mask, size_t m_nb1, bool mask_f16, const float * sinks, float * dst, size_t dst_nb1,
                 int64_t head_dim, int64_t n_kv, int64_t n_head, int64_t n_head_kv, int64_t T, float scale, hipStream_t stream) {
    attn_args a = { (const char *) q, q_nb1, q_nb2, (const char *) k, k_nb1, k_nb2, (const char *) v, v_nb1, v_nb2,
                    (const char *) mask, m_nb1, mask_f16 ? 1 : 0, sinks, dst, dst_nb1, (int) n_kv, (int) n_head, (int) n_head_kv, (int) T, scale };
    const dim3 grid((unsigned) n_head, (unsigned) T);
    const size_t lds = (size_t) n_kv*4;
    if (head_dim == 128) hipLaunchKernelGGL((k_attn_decode<128>), grid, dim3(256), lds, stream, a);
    else                 hipLaunchKernelGGL((k_attn_decode<64>),  grid, dim3(256), lds, stream, a);
}

// ---------------------------------------------------------------------------------------------------------------
// grouped mat-vec with epilogues (n = 1)
// ---------------------------------------------------------------------------------------------------------------
struct fused_rope {
    const int32_t * pos; const float * ff; int n_dims, head_dim, n_ctx_orig;
    float freq_scale, ext_factor, attn_factor, theta_scale, corr_lo, corr_hi;
};

struct fused_mmvq_args {
    mmvq_group g[MMVQ_MAX_GROUPS];
    int n_groups;
    int block_end[MMVQ_MAX_GROUPS];       // cumulative workgroup counts
    int k;
    int act_kind;
    const int8_t * a_qs; const float * a_d; const int16_t * a_bs;   // quantized activation (global), or:
    const float * x_f32;                                              // f32 activation to quantize in the prologue (a_qs == null)
    fused_rope rope;
};

static __device__ __forceinline__ void rope_pair(const fused_rope & r, int row_in_head, float & x0, float & x1) {
    // NORM rope on the pair (2i, 2i+1) — same formulas as elem.hip k_rope<false>
    if (row_in_head >= r.n_dims) return;
    const int ip = row_in_head >> 1;
    const float theta_base = (float) r.pos[0]*powf(r.theta_scale, (float) ip);
    const float theta_extrap = theta_base/(r.ff ? r.ff[ip] : 1.0f);
    float theta_interp = r.freq_scale*theta_extrap, theta = theta_interp, mscale = r.attn_factor;
    if (r.ext_factor != 0.0f) {
        const float y = ((float) ip - r.corr_lo)/fmaxf(0.001f, r.corr_hi - r.corr_lo);
        const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, y)))*r.ext_factor;
        theta = theta_interp*(1.0f - ramp_mix) + theta_extrap*ramp_mix;
        mscale *= 1.0f + 0.1f*logf(1.0f/r.freq_scale);
    }
    const float c = cosf(theta)*mscale, s = sinf(theta)*mscale;
    const float a = x0, b = x1;
    x0 = a*c - b*s;
    x1 = a*s + b*c;
}

template <int TYPE>
static __device__ __forceinline__ void fused_group_rows(const mmvq_group & g, const fused_mmvq_args & p, const act_view & av, int blk_in_group,
                                                        int lane, int wave) {
    constexpr int R = 2;
    const int64_t nb = p.k / mmvq_t<TYPE>::QK;
    const int row0 = (blk_in_group*4 + wave)*R;
    if (row0 >= g.m) return;
    const char * rows[R];
#pragma unroll
    for (int r = 0; r < R; r++) rows[r] = g.W + (size_t) min(row0 + r, g.m - 1)*g.row_stride;
    const act_view avs[1] = { av };
    float acc[1][R] = { { 0.0f, 0.0f } };
    mmvq_wave_partial<TYPE, 1, R>(rows, avs, nb, lane, acc);
    float s0 = wave_sum(acc[0][0]), s1 = wave_sum(acc[0][1]);
    if (g.epi == EPI_GLU) {
        const char * rows2[R];
#pragma unroll
        for (int r = 0; r < R; r++) rows2[r] = g.W2 + (size_t) min(row0 + r, g.m - 1)*g.row_stride;
        float acc2[1][R] = { { 0.0f, 0.0f } };
        mmvq_wave_partial<TYPE, 1, R>(rows2, avs, nb, lane, acc2);
        const float u0 = wave_sum(acc2[0][0]), u1 = wave_sum(acc2[0][1]);
        s0 = (s0/(1.0f + expf(-s0)))*u0;      // silu(gate)*up, as elem.hip k_glu
        s1 = (s1/(1.0f + expf(-s1)))*u1;
    }
    if (lane != 0) return;
    if (g.epi == EPI_ADD) {
        s0 += g.res[row0];
        if (row0 + 1 < g.m) s1 += g.res[row0 + 1];
    } else if (g.epi == EPI_ROPE) {
        rope_pair(p.rope, row0 % p.rope.head_dim, s0, s1);   // m is even on this path
    }
    g.dst[row0] = s0;
    if (row0 + 1 < g.m) g.dst[row0 + 1] = s1;
}

__global__ void __launch_bounds__(256) k_mmvq_fused(const fused_mmvq_args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    act_view av;
    if (p.a_qs) {
        if (p.act_kind == T_Q8_0) av = stage_act_lds<T_Q8_0>(smem, p.a_qs, p.a_d, p.a_bs, p.k);
        else                      av = stage_act_lds<T_Q8_K>(smem, p.a_qs, p.a_d, p.a_bs, p.k);
    } else {
        // prologue quantization of the f32 activation vector straight into LDS (k % 256 == 0)
        const int nd = p.act_kind == T_Q8_0 ? 32 : 256, nbs = p.act_kind == T_Q8_0 ? 32 : 16;
        const int64_t qs_b = (p.k + 15) & ~15, d_b = ((p.k/nd)*4 + 15) & ~15;
        int8_t * qs = (int8_t *) smem; float * d = (float *) (smem + qs_b); int16_t * bs = (int16_t *) (smem + qs_b + d_b);
        for (int c = wave; c < p.k/256; c += 4) {
            const float4v v = *(const float4v *) (p.x_f32 + c*256 + lane*4);
            if (p.act_kind == T_Q8_0) quant_store_chunk256<T_Q8_0>(v, c, lane, qs, d, bs);
            else                      quant_store_chunk256<T_Q8_K>(v, c, lane, qs, d, bs);
        }
        av.qs = qs; av.d = d; av.bs = bs;
    }
    __syncthreads();

    int gi = 0;
    while (gi < p.n_groups - 1 && (int) blockIdx.x >= p.block_end[gi]) gi++;
    const int blk = (int) blockIdx.x - (gi ? p.block_end[gi - 1] : 0);
    const mmvq_group & g = p.g[gi];
    switch (g.type) {
        case T_Q4_K:  fused_group_rows<T_Q4_K >(g, p, av, blk, lane, wave); break;
        case T_Q6_K:  fused_group_rows<T_Q6_K >(g, p, av, blk, lane, wave); break;
        case T_Q5_K:  fused_group_rows<T_Q5_K >(g, p, av, blk, lane, wave); break;
        case T_Q8_0:  fused_group_rows<T_Q8_0 >(g, p, av, blk, lane, wave); break;
        case T_Q4_0:  fused_group_rows<T_Q4_0 >(g, p, av, blk, lane, wave); break;
        case T_MXFP4: fused_group_rows<T_MXFP4>(g, p, av, blk, lane, wave); break;
        default: break;
    }
}

static float rope_corr_dim_h(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims*logf(n_ctx_orig/(n_rot*2*(float) M_PI))/(2*logf(base));
}

void mul_mat_vec_q_fused(const mmvq_group * groups, int n_groups, int64_t k, const act_q8 * act, const float * x_f32, int act_kind,
                         const mmvq_rope * rope, hipStream_t stream) {
    fused_mmvq_args a = {};
    a.n_groups = n_groups; a.k = (int) k; a.act_kind = act_kind;
    int blocks = 0;
    for (int i = 0; i < n_groups; i++) {
        a.g[i] = groups[i];
        blocks += (int)((groups[i].m + 7)/8);
        a.block_end[i] = blocks;
    }
    if (act) { a.a_qs = act->qs; a.a_d = act->d; a.a_bs = act->bsums; } else { a.x_f32 = x_f32; }
    if (rope) {
        a.rope.pos = rope->pos; a.rope.ff = rope->freq_factors; a.rope.n_dims = rope->p.n_dims; a.rope.head_dim = rope->head_dim;
        a.rope.n_ctx_orig = rope->p.n_ctx_orig; a.rope.freq_scale = rope->p.freq_scale; a.rope.ext_factor = rope->p.ext_factor;
        a.rope.attn_factor = rope->p.attn_factor;
        a.rope.theta_scale = powf(rope->p.freq_base, -2.0f/rope->p.n_dims);
        const float start = floorf(rope_corr_dim_h(rope->p.n_dims, rope->p.n_ctx_orig, rope->p.beta_fast, rope->p.freq_base));
        const float end   = ceilf (rope_corr_dim_h(rope->p.n_dims, rope->p.n_ctx_orig, rope->p.beta_slow, rope->p.freq_base));
        a.rope.corr_lo = fmaxf(0.0f, start); a.rope.corr_hi = fminf((float)(rope->p.n_dims - 1), end);
    }
    hipLaunchKernelGGL(k_mmvq_fused, dim3((unsigned) blocks), dim3(256), act_lds_bytes(act_kind, k), stream, a);
}

bool mul_mat_vec_q_fused_supported(int64_t k, int act_kind) { return k % 256 == 0 && act_lds_bytes(act_kind, k) <= 64*1024; }

} // namespace mi355x
